#!/bin/bash
# per-kernel averages of the bench at another size:  tools/kstats_n.sh 100000 [steps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-100000}; K=${2:-60}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_ksn
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ksn -o ks -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference --floes $N --steps $K --warmup 10 > $R/gpurun_out/prof_ksn.json 2> $R/gpurun_out/prof_ksn.err
cut -c1-160 $R/gpurun_out/prof_ksn.json
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_ksn/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'].split('(')[0][-44:]:46s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us")
PY
