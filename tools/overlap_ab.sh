#!/bin/bash
# experiment: forcings on a second stream beside the collision kernels (SZ_OVERLAP=1)
for n in 10000 40000 100000; do for o in 0 1; do
  SZ_OVERLAP=$o python bench.py --floes $n --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('overlap=$o n=$n ms/step %.4f'%j['ms_per_step'], 'narrow %.4f'%j['roofline']['kernel_ms'])"
done; done
