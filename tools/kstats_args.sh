#!/bin/bash
# per-kernel averages of the bench with extra arguments:  tools/kstats_args.sh --two-way --steps 60
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_ksa
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ksa -o ks -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference "$@" > $R/gpurun_out/prof_ksa.json 2> $R/gpurun_out/prof_ksa.err
cut -c1-160 $R/gpurun_out/prof_ksa.json
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_ksa/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f"{r['Name'].split('(')[0][-44:]:46s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us")
PY
