#!/bin/bash
# experiment: resident workgroups of the first narrow variant (work-queue kernel)
for g in 0 2048; do
  for n in 10000 20000 40000; do
  SZ_NARROW_GRID=$g python bench.py --floes $n --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_step']; print('grid=$g n=$n narrow_ms %.4f'%j['roofline']['kernel_ms'], 'large %.4f'%k['narrow_large'], 'ms/step %.4f'%j['ms_per_step'])"
  done
done
