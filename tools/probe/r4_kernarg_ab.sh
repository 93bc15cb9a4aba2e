# where the kernel arguments live (host or device memory): the narrow launch reads a 2 KB State by value
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["ms_per_step_min"],4), d["roofline"].get("pipelined_steps"), {k: round(v,4) for k,v in d["kernel_ms_per_step"].items() if v})'
for v in "SZ_X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "SZ_X=0" ; do echo $v; env $v python bench.py --no-cpu-baseline --no-strong-reference --repeats 5 2>/dev/null | python -c "$P"; done
