#!/bin/bash
# the round's closing measurements on the final sources (run from the repo root on the GPU box)
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["ms_per_step_min"],4), round(d["value"]/1e6,1), "M floe-steps/s", d["roofline"].get("pipelined_steps"))'
tools/profile_round.sh r04 > gpurun_out/r4_profile_10k.log 2>&1; tail -2 gpurun_out/r4_profile_10k.log
tools/profile_round_100k.sh r04_100k > gpurun_out/r4_profile_100k.log 2>&1; tail -2 gpurun_out/r4_profile_100k.log
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; python -c "$P" < gpurun_out/r04_bench.json
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_flags.json 2> gpurun_out/r04_bench_driver_flags.err; python -c "$P" < gpurun_out/r04_bench_driver_flags.json
{
echo "configs[3] 10000 floes between walls + topography:"; python bench.py --no-cpu-baseline --no-strong-reference --workload configs3 --repeats 5 2>/dev/null | python -c "$P"
} > gpurun_out/r4_walls.txt 2>&1; cat gpurun_out/r4_walls.txt
{
for n in 20000 40000; do echo "n=$n:"; python bench.py --no-cpu-baseline --no-strong-reference --floes $n --repeats 5 2>/dev/null | python -c "$P"; done
echo "forced tiled 12.5k:"; python bench.py --no-cpu-baseline --no-strong-reference --force-tiled --floes 12500 --repeats 5 2>/dev/null | python -c "$P"
echo "single 12.5k:"; python bench.py --no-cpu-baseline --no-strong-reference --floes 12500 --repeats 5 2>/dev/null | python -c "$P"
echo "driver flags x3 (batched clears):"; for i in 1 2 3; do python bench.py --no-cpu-baseline --no-strong-reference --steps 20 --warmup 5 2>/dev/null | python -c "$P"; done
} > gpurun_out/r4_final_sweep.txt 2>&1; cat gpurun_out/r4_final_sweep.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python -m pytest tests -x -q -m gpu > gpurun_out/r4_final_gpu_suite.txt 2>&1; tail -2 gpurun_out/r4_final_gpu_suite.txt
