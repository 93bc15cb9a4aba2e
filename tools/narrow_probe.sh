#!/bin/bash
# timing experiment: narrow-phase time with parts of the kernel disabled (results are wrong on purpose)
N=${1:-10000}
for d in 0 2 1 4; do
  SZ_DEBUG=$d python bench.py --floes $N --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('N=$N SZ_DEBUG=$d narrow_ms', j['roofline']['kernel_ms'], 'ms/step', j['ms_per_step'])"
done
