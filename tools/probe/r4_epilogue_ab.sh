P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["ms_per_step_min"],4), {k: round(v,4) for k,v in d["kernel_ms_per_step"].items() if v})'
for v in "SZ_X=0" "SZ_DEBUG=32" "SZ_X=0" "SZ_DEBUG=32"; do echo $v; env $v python bench.py --no-cpu-baseline --no-strong-reference --repeats 5 2>/dev/null | python -c "$P"; done
