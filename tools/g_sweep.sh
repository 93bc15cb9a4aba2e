#!/bin/bash
# experiment: lanes per pair in the narrow phase
for g in 16 8; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DNARROW_G=$g -o subzero.jl_amd/libsubzero_hip.so subzero.jl_amd/csrc/sz_api.hip 2>/dev/null
  for n in 5000 10000 40000; do
  python bench.py --floes $n --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('G=$g n=$n narrow_ms %.4f'%j['roofline']['kernel_ms'], 'ms/step %.4f'%j['ms_per_step'])"
  done
done
