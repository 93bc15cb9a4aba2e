#!/bin/bash
# what the first launch of a pipelined step (narrow | GEO | forcing tail) is made of: kernel durations with the clips cut off (SZ_DEBUG=4: staging only) and without coupling
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # env..., then bench args after --
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for kv in "${envs[@]}"; do export "$kv"; done
  cd /tmp && export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_lp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lp -o ks -- python3 $R/bench.py --no-cpu-baseline --no-strong-reference --repeats 2 "$@" > /dev/null 2> $R/gpurun_out/prof_lp.err
  python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_lp/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:2]:
    print(f"   {r['Name'].split('(')[0][-44:]:46s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us")
PY
  for kv in "${envs[@]}"; do unset "${kv%%=*}"; done
  cd $R
}
echo "full step:";                          run SZ_DEBUG=0 --
echo "no coupling (narrow | GEO):";         run SZ_DEBUG=0 -- --coupling-dt 1000000
echo "staging only + forcing + GEO:";       run SZ_DEBUG=4 --
echo "staging only, no coupling (GEO):";    run SZ_DEBUG=4 -- --coupling-dt 1000000
