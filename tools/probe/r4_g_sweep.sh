# lanes per item of the first narrow variant (NARROW_G): A/B builds on the GPU box
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["ms_per_step_min"],4), {k: round(v,4) for k,v in d["kernel_ms_per_step"].items() if v})'
for f in "-DNARROW_G=4" "-DNARROW_G=16" "-DNARROW_G=8"; do
  echo "build $f"; SZ_EXTRA_FLAGS="$f" python subzero.jl_amd/build.py > gpurun_out/g_build.log 2>&1 || { echo build failed; grep -m3 error gpurun_out/g_build.log; continue; }
  python tools/kernel_resources.py "sz_k_narrow<4" "sz_k_narrow<8, 18" "sz_k_narrow<16, 18" | cut -c1-130 | head -3
  echo " 10k:"; python bench.py --no-cpu-baseline --no-strong-reference --repeats 5 2>/dev/null | python -c "$P"
  echo " 40k:"; python bench.py --no-cpu-baseline --no-strong-reference --floes 40000 --repeats 3 2>/dev/null | python -c "$P"
  echo " 100k:"; python bench.py --no-cpu-baseline --no-strong-reference --floes 100000 --workload configs2 --steps 50 --repeats 5 2>/dev/null | python -c "$P"
done
