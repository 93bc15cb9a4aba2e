#!/bin/bash
# effective shader clock during the bench kernels: GRBM_GUI_ACTIVE (cycles the GPU was busy, per launch) against the
# launch duration from the kernel trace of the same run:  tools/clock_probe.sh [n_floes]
R=${GRAFT_REPO_ROOT:-$(pwd)}; N=${1:-10000}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_clk
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_clk -o c -- python3 $R/bench.py --no-cpu-baseline --floes $N --steps 20 --warmup 5 > /dev/null 2> $R/gpurun_out/prof_clk.err
cd $R && python3 - <<PY
import csv, glob
from collections import defaultdict
dur = {}
for f in glob.glob("gpurun_out/prof_clk/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = defaultdict(lambda: [0, 0.0, 0.0])
for f in glob.glob("gpurun_out/prof_clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur: continue
        name, d = dur[r["Dispatch_Id"]]
        a = acc[name.split("(")[0][-40:]]; a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += d
for k, (n, cyc, ns) in sorted(acc.items(), key=lambda kv: -kv[1][2])[:12]:
    print(f"{k:42s} {n:5d} launches  {ns / n / 1e3:8.1f} us  {cyc / n:12.0f} cycles  -> {cyc / ns * 1e3:7.0f} MHz (if the counter is per device)")
PY
