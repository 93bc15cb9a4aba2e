# the one-way forcing kernel ALONE on the chip (100 k floes, SZ_OVERLAP=0 SZ_FUSE_FORCING=0: own launch in the chain), event-timed class "forcing"
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), {k: round(v,4) for k,v in d["kernel_ms_per_step"].items() if v})'
for v in "SZ_BLOCK_POINTS=1" "SZ_BLOCK_POINTS=0"; do echo $v; env $v SZ_OVERLAP=0 SZ_FUSE_FORCING=0 python bench.py --no-cpu-baseline --no-strong-reference --floes 100000 --workload configs2 --steps 50 --repeats 3 2>/dev/null | python -c "$P"; done
for v in "SZ_BLOCK_POINTS=1" "SZ_BLOCK_POINTS=0"; do echo "10k $v"; env $v SZ_PIPE_MAX_FLOES=0 SZ_OVERLAP=0 SZ_FUSE_FORCING=0 python bench.py --no-cpu-baseline --no-strong-reference --repeats 3 2>/dev/null | python -c "$P"; done
