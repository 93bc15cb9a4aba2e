#!/bin/bash
# experiment: dynamic rounds (8 queue heads) vs static split in the narrow phase
for n in 10000 20000 40000 100000; do for q in 0 1; do
  SZ_NARROW_QUEUE=$q python bench.py --floes $n --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('queue=$q n=$n ms/step %.4f'%j['ms_per_step'], 'narrow %.4f'%j['roofline']['kernel_ms'])"
done; done
