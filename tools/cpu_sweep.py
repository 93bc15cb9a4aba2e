import os, sys, time
sys.path.insert(0, os.getcwd())
from subzero_jl_amd import fields
from oracle import orc
cfg = fields.make_config(n_floes=10000, seed=12345)
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cpu.max", e)
for nt in (1, 8, 16, 32, 64, 128):
    w = fields.build_world(orc.World(), cfg); w.set_threads(nt)
    w.timestep_sim(0, cfg["dt"], coupling_dt=1)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 4 and n < 20:
        w.timestep_sim(1 + n, cfg["dt"], coupling_dt=1); n += 1
    el = time.perf_counter() - t0
    print(nt, "threads:", 10000 * n / el, "floe-steps/s", n, "steps")
