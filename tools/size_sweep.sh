#!/bin/bash
# narrow-phase kernel time vs field size (pairs in flight vs chain latency)
for n in ${@:-500 2000 5000 10000 20000 40000 100000 200000}; do
  python bench.py --floes $n --steps 100 --warmup 10 --repeats 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_step']; print('n=$n pairs', j['counts']['n_pairs'], 'narrow_ms %.4f'%j['roofline']['kernel_ms'], 'ms/step %.4f'%j['ms_per_step'], 'Mfs/s %.2f'%(j['value']/1e6), {a:round(b,4) for a,b in k.items()})"
done
