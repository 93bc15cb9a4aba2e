P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["ms_per_step_min"],4), d["roofline"].get("pipelined_steps"))'
for n in 20000 40000; do for pm in 1000000 0; do echo "n=$n pipe_max=$pm"; SZ_PIPE_MAX_FLOES=$pm python bench.py --no-cpu-baseline --no-strong-reference --floes $n --repeats 5 2>/dev/null | python -c "$P"; done; done
echo "100k configs2:"; for v in "SZ_PIPE_MAX_FLOES=0" "SZ_PIPE_MAX_FLOES=1000000" "SZ_PIPE_MAX_FLOES=1000000 SZ_OVERLAP=0"; do echo $v; env $v python bench.py --no-cpu-baseline --no-strong-reference --floes 100000 --workload configs2 --steps 50 --repeats 5 2>/dev/null | python -c "$P"; done
echo "forced tiled 12.5k:"; python bench.py --no-cpu-baseline --no-strong-reference --force-tiled --floes 12500 --repeats 5 2>/dev/null | python -c "$P"
echo "single 12.5k:"; python bench.py --no-cpu-baseline --no-strong-reference --floes 12500 --repeats 5 2>/dev/null | python -c "$P"
