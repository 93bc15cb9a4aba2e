#!/bin/bash
# kernel durations at 100 k floes, every kernel alone on the chip (SZ_OVERLAP=0), under extra environment settings:  r4_kstats_100k_env.sh SZ_REDUCE_FREE=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_ka
SZ_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ka -o ks -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference --floes 100000 --workload configs2 --steps 50 --repeats 3 2> $R/gpurun_out/prof_ka.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],4))"
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_ka/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(f"{r['Name'].split('(')[0][-44:]:46s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us")
PY
