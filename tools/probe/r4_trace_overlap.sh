#!/bin/bash
# kernel trace of a few pipelined steps with the forcings on the second stream: do the two kernels really run at the same time?
R=${GRAFT_REPO_ROOT:-$(pwd)}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tr
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tr -o tr -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference --steps 40 --warmup 10 --repeats 2 > /dev/null 2> $R/gpurun_out/prof_tr.err
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
k = len(rows) - 40
t0 = int(rows[k]["Start_Timestamp"])
for r in rows[k:k + 24]:
    print(f"{r['Kernel_Name'].split('(')[0][-40:]:42s} queue {r.get('Queue_Id','?'):>3s}  start {(int(r['Start_Timestamp']) - t0) / 1e3:8.2f}  end {(int(r['End_Timestamp']) - t0) / 1e3:8.2f}  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.2f} us")
PY
