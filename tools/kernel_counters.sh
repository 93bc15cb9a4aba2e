#!/bin/bash
# instruction / memory counters of ONE kernel (name substring), averages per launch, separate --pmc passes:
#   tools/kernel_counters.sh sz_k_integrate 100000
R=${GRAFT_REPO_ROOT:-$(pwd)}; K=${1:-sz_k_integrate}; N=${2:-10000}
cd /tmp && export TMPDIR=/tmp
i=0
rm -rf $R/gpurun_out/prof_kc*
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/prof_kc$i -o s -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference --floes $N --steps 10 --warmup 3 > /dev/null 2> $R/gpurun_out/prof_kc$i.err || { tail -3 $R/gpurun_out/prof_kc$i.err; continue; }
done
cd $R && python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/prof_kc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$K" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"{k:36s} {v / n:16.1f}  ({n} launches)")
PY
