# the one-launch integrator compiled for one / two wavefronts per SIMD (both with the moved ring read back in the ghost and pack tails)
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["ms_per_step_min"],4), {k: round(v,4) for k,v in d["kernel_ms_per_step"].items() if v})'
for f in "-DSZ_INTEG_WPE=2" "-DSZ_INTEG_WPE=1"; do
  echo "build $f"; SZ_EXTRA_FLAGS="$f" python subzero.jl_amd/build.py > /dev/null 2>&1 || { echo build failed; continue; }
  echo " forced tiled 12.5k:"; python bench.py --no-cpu-baseline --no-strong-reference --force-tiled --floes 12500 --repeats 5 2>/dev/null | python -c "$P"
  echo " 10k three-launch (SZ_PIPELINE=0):"; SZ_PIPELINE=0 python bench.py --no-cpu-baseline --no-strong-reference --repeats 5 2>/dev/null | python -c "$P"
  echo " 100k:"; python bench.py --no-cpu-baseline --no-strong-reference --floes 100000 --workload configs2 --steps 50 --repeats 5 2>/dev/null | python -c "$P"
done
