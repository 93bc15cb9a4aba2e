#!/bin/bash
# timeline of two steps at 100 k floes (three-launch steps, forcings on the second stream)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_t1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_t1 -o t -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference --floes 100000 --workload configs2 --steps 30 --warmup 5 --repeats 2 > /dev/null 2> $R/gpurun_out/prof_t1.err
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_t1/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
k = len(rows) - 60
t0 = int(rows[k]["Start_Timestamp"])
for r in rows[k:k + 14]:
    print(f"{r['Kernel_Name'].split('(')[0][-40:]:42s} q{r.get('Queue_Id','?'):>2s}  start {(int(r['Start_Timestamp']) - t0) / 1e3:8.2f}  end {(int(r['End_Timestamp']) - t0) / 1e3:8.2f}  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.2f}")
PY
