#!/bin/bash
# the other BASELINE.json configs on one GPU (parity-test cases; numbers for DESIGN.md, not bench lines)
for w in "configs1 10000" "configs3 10000" "configs1 100000" "configs4 100000"; do
  set -- $w
  python bench.py --workload $1 --floes $2 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_step']; c=j['counts']; print('$1 n=$2 ms/step %.4f'%j['ms_per_step'], 'Mfs/s %.2f'%(j['value']/1e6), 'narrow_ms %.4f'%j['roofline']['kernel_ms'], 'frac %.4f'%j['roofline']['frac'], 'pairs', c['n_pairs'], 'elem_rows', c['n_elem_rows'], 'ghosts', c['n_ghosts'], 'retry', c['n_retry'], {a:round(b,4) for a,b in k.items()})"
done
