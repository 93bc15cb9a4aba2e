#!/bin/bash
# instruction mix / issue utilisation of the narrow kernel (separate --pmc passes), averages per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; N=${1:-10000}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU" "SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/prof_sq$i -o s -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference --floes $N --steps 10 --warmup 3 > /dev/null 2> $R/gpurun_out/prof_sq$i.err || exit 1
done
cd $R && python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/prof_sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sz_k_narrow<8" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"{k:32s} {v / n:16.1f}  ({n} launches)")
PY
