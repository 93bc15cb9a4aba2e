#!/bin/bash
# what the one-way forcing kernel (own launch: fields above 65 536 floes, or SZ_OVERLAP=1) waits for -- issue, vector-memory and texture-path counters,
# averages per launch, separate --pmc passes:   tools/frc_counters.sh [n_floes] [workload]
R=${GRAFT_REPO_ROOT:-$(pwd)}; N=${1:-100000}; W=${2:-configs2}; K=sz_k_forcing
cd /tmp && export TMPDIR=/tmp
i=0
rm -rf $R/gpurun_out/prof_fc*
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS"; do
  i=$((i+1))
  # (a counter set the hardware cannot collect in one pass makes rocprofv3 abort and then hang in its finalisation: every pass under a timeout)
  SZ_OVERLAP=1 timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/prof_fc$i -o s -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference --floes $N --workload $W --steps 10 --warmup 3 --repeats 2 > /dev/null 2> $R/gpurun_out/prof_fc$i.err || { tail -3 $R/gpurun_out/prof_fc$i.err; continue; }
done
cd $R && python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/prof_fc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$K" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"{k:40s} {v / n:16.1f}  ({n} launches)")
PY
