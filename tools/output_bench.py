"""Cost of the output path on the resident state: write_grid_data (ghosts on, calc_eulerian_data!, ghosts off) and
simplify_check at the bench workload, next to the oracle's calc_eulerian_data on the host cores.
usage: python tools/output_bench.py [n_floes]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subzero_jl_amd
from subzero_jl_amd import fields
from oracle import orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfg = fields.make_config(n_floes=n, seed=12345, ocean="converge_diverge")
hw = fields.build_world(subzero_jl_amd.World(0), cfg)
ow = fields.build_world(orc.World(), cfg)
hw.run(20, 0, cfg["dt"], coupling_dt=10)
for t in range(20):
    ow.timestep_sim(t, cfg["dt"], coupling_dt=10)
L = cfg["L"]
for dims in ((10, 10), (100, 100), (400, 400)):
    xg, yg = np.linspace(0, L, dims[0] + 1), np.linspace(0, L, dims[1] + 1)
    hw.write_grid_data(xg, yg)
    t0 = time.perf_counter()
    for _ in range(5):
        got = hw.write_grid_data(xg, yg)
    tg = (time.perf_counter() - t0) / 5
    n0 = ow.M
    ow.add_ghosts()
    t0 = time.perf_counter()
    ref = ow.eulerian_data(xg, yg)
    tc = time.perf_counter() - t0
    ow.remove_ghosts(n0)
    errs = [np.abs(got[k] - ref[k]).max() / max(np.abs(ref[k]).max(), 1e-300) for k in range(len(ref))]
    worst = int(np.argmax(errs))
    print(f"floes {n} grid {dims[0]}x{dims[1]}: hip {tg * 1e3:.2f} ms  oracle {tc * 1e3:.1f} ms  max diff / max|ref| {max(errs):.1e} ({hw.EUL_OUTPUTS[worst]}; "
          f"area {errs[6]:.1e}, si_frac {errs[8]:.1e}, u {errs[0]:.1e})", flush=True)
t0 = time.perf_counter()
for _ in range(20):
    s = hw.simplify_check(30, 1e6, 0.1)
print(f"simplify_check: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", s)
