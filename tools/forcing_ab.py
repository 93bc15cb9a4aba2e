#!/usr/bin/env python3
"""A/B of the stand-alone one-way forcing launch (timestep_coupling!, coupling.jl:1486-1589) on one field:
    python3 tools/forcing_ab.py [n_floes] [precision]
SZ_FRC_X = 0: forcing_body; 7: the same loop as a template (control); 1: next trip's point prefetched; 2: leaner arithmetic (fused rotation,
four bilinear weights, sqrt by rsq + two Newton steps); 3: both; 4: both with 16 lanes per floe; 5: both, compiled for 5 wavefronts per SIMD;
6: lean, 16 lanes, 5 wavefronts per SIMD.
Prints the event-timed kernel time of each variant and whether the four forcing columns are bit-equal to variant 0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import subzero_jl_amd
from subzero_jl_amd import fields

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
prec = sys.argv[2] if len(sys.argv) > 2 else "f64"
wl = dict(seed=12346, ocean="converge_diverge") if n >= 50000 else dict(seed=12345)
cfg = fields.make_config(n_floes=n, **wl)
ref = None
for var in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("0", "7", "1", "2", "3", "4", "5", "6")):
    os.environ["SZ_FRC_X"] = var
    w = fields.build_world(subzero_jl_amd.World(0), cfg); w.set_precision(prec)
    w.run(3, 0, cfg["dt"], coupling_dt=1, stop_on_tags=False)          # a few steps: rotated floes, wrapped parents
    w.timestep_coupling()
    w.profile(True, only="forcing")
    for _ in range(20):
        w.timestep_coupling()
    ms, k = w.kernel_times()["forcing"]
    out = [w.get(f) for f in ("fxOA", "fyOA", "trqOA", "hflx_factor")]
    if ref is None:
        ref = out
    eq = all(np.array_equal(a, b) for a, b in zip(out, ref))
    rel = max(float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)) for a, b in zip(out, ref))
    st = w.stats()
    byts = 16 * st["n_sub_points"] + 96 * n
    print(f"SZ_FRC_X={var} {prec}: {1e3 * ms / k:8.1f} us per launch ({k} launches), {byts / (ms / k * 1e-3) / 1e12:.2f} TB/s algorithmic, bit-equal to variant 0: {eq}, max rel diff {rel:.2e}", flush=True)
    del w
