#!/bin/bash
# The PMC passes of tools/profile_round.sh for the metric's other size: configs[2], 100 000 floes in one context (kernel trace + stats, then one
# --pmc pass each for FETCH_SIZE and WRITE_SIZE); merged into gpurun_out/pmc_traffic.json under "configs2:100000".
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03_100k}
A="--no-cpu-baseline --no-strong-reference --floes 100000 --workload configs2"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_kt $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -o kt -- python3 $R/bench.py $A --steps 50 --repeats 3 > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/prof_kt.err &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o f -- python3 $R/bench.py $A --steps 20 --warmup 5 --repeats 1 > /dev/null 2> $R/gpurun_out/prof_f.err &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o w -- python3 $R/bench.py $A --steps 20 --warmup 5 --repeats 1 > /dev/null 2> $R/gpurun_out/prof_w.err &&
cd $R && python3 tools/pmc_summary.py gpurun_out ${TAG} configs2:100000
