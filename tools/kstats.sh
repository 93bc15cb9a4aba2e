#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of the default bench under an environment setting:
#   tools/kstats.sh SZ_STATIC_GRID=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_ks
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ks -o ks -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference > /dev/null 2> $R/gpurun_out/prof_ks.err
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_ks/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'].split('(')[0][-44:]:46s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us")
PY
