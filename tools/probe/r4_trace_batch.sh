#!/bin/bash
# what lies between two 20-step batches (the driver's flags): every launch from the last step of one batch to the first step of the next
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tb
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/prof_tb -o tb -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference --steps 20 --warmup 5 --repeats 6 > /dev/null 2> $R/gpurun_out/prof_tb.err
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$R/gpurun_out/prof_tb/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]))
for f in glob.glob("$R/gpurun_out/prof_tb/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
# the last batch boundary: find the last sz_k_inter_fill and print 14 before .. 22 after
idx = [i for i, r in enumerate(rows) if "inter_fill" in r[2]]
k = idx[-2] if len(idx) > 1 else idx[-1]
t0 = rows[k - 6][0]
for s, e, n in rows[k - 6:k + 26]:
    print(f"{n:46s} start {(s - t0) / 1e3:9.2f}  dur {(e - s) / 1e3:7.2f} us")
PY
