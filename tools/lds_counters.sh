#!/bin/bash
# LDS side of the narrow kernel (separate --pmc passes): instructions, bank / address conflicts, busy and wait cycles, averages per launch
#   tools/lds_counters.sh [n_floes]
R=${GRAFT_REPO_ROOT:-$(pwd)}; N=${1:-100000}
cd /tmp && export TMPDIR=/tmp
i=0
rm -rf $R/gpurun_out/prof_lds*
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/prof_lds$i -o s -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference --floes $N --workload configs2 --steps 10 --warmup 3 --repeats 2 > /dev/null 2> $R/gpurun_out/prof_lds$i.err || { tail -3 $R/gpurun_out/prof_lds$i.err; continue; }
done
cd $R && python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/prof_lds*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sz_k_narrow<8" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"{k:32s} {v / n:16.1f}  ({n} launches)")
PY
