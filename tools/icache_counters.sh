#!/bin/bash
# instruction-cache and issue counters of the step's kernels (separate --pmc passes), averages per launch:
#   tools/icache_counters.sh [n_floes] [ENV=VALUE ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; N=${1:-10000}; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_ic*
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_MISSES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/prof_ic$i -o s -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference --floes $N --steps 10 --warmup 3 --repeats 2 > /dev/null 2> $R/gpurun_out/prof_ic$i.err || exit 1
done
cd $R && python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/prof_ic*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void sz::", "").replace("sz::", "")
        if not k.startswith("sz_k_"): continue
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", [0, 0])[1])[:5]:
    print(k)
    for c, (n, v) in sorted(acc[k].items()):
        print(f"   {c:32s} {v / n:16.1f}  ({n} launches)")
PY
