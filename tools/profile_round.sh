#!/bin/bash
# The three rocprofv3 passes behind the numbers in bench.py / DESIGN.md (run on the GPU box from the repo root):
#   kernel trace + stats, then one --pmc pass each for FETCH_SIZE and WRITE_SIZE (separate passes, as the
#   MI355X guide prescribes).  Summaries land in gpurun_out/; copy the ones to keep into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_kt $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference --repeats 2 > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/prof_kt.err &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference --steps 40 --warmup 5 --repeats 1 > /dev/null 2> $R/gpurun_out/prof_f.err &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o w -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference --steps 40 --warmup 5 --repeats 1 > /dev/null 2> $R/gpurun_out/prof_w.err &&
cd $R && python3 tools/pmc_summary.py gpurun_out ${TAG}
