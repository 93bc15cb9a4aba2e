# quick step-level numbers: 10 k (200-step blocks and the driver's flags), 100 k, forced-tiled 12.5 k
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["ms_per_step_min"],4), d["roofline"].get("pipelined_steps"), {k: round(v,4) for k,v in d["kernel_ms_per_step"].items() if v})'
echo "10k:"; python bench.py --no-cpu-baseline --no-strong-reference --repeats 5 2>/dev/null | python -c "$P"
echo "10k driver flags:"; python bench.py --no-cpu-baseline --no-strong-reference --steps 20 --warmup 5 2>/dev/null | python -c "$P"
echo "100k configs2:"; python bench.py --no-cpu-baseline --no-strong-reference --floes 100000 --workload configs2 --steps 50 --repeats 5 2>/dev/null | python -c "$P"
echo "forced tiled 12.5k:"; python bench.py --no-cpu-baseline --no-strong-reference --force-tiled --floes 12500 --repeats 5 2>/dev/null | python -c "$P"
