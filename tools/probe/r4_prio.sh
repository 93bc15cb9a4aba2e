P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["ms_per_step_min"],4), {k: round(v,4) for k,v in d["kernel_ms_per_step"].items() if v})'
for f in "-DSZ_NARROW_PRIO=3" "-DSZ_X=0" "-DSZ_NARROW_PRIO=1"; do
  echo "build $f"; SZ_EXTRA_FLAGS="$f" python subzero.jl_amd/build.py > /dev/null 2>&1 || { echo build failed; continue; }
  for i in 1 2; do python bench.py --no-cpu-baseline --no-strong-reference --repeats 5 2>/dev/null | python -c "$P"; done
  echo " 100k:"; python bench.py --no-cpu-baseline --no-strong-reference --floes 100000 --workload configs2 --steps 50 --repeats 5 2>/dev/null | python -c "$P"
done
