"""two-way coupling: the rectangle pipeline (default) against the general clipper (SZ_TW_GENERAL_CLIP=1) on seeded
fields -- ocean stress, sea-ice fraction and trajectories after a few coupled steps.  usage: python tools/tw_ab.py [nseeds]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subzero_jl_amd
from subzero_jl_amd import fields

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
worst = 0.0
for seed in range(nseeds):
    for walls, conc, n in ((False, 0.8, 1500), (True, 0.8, 900), (False, 0.45, 1200), (True, 0.95, 700)):
        cfg = fields.make_config(n_floes=n, seed=100 + seed, walls=walls, concentration=conc,
                                 ocean="converge_diverge" if not walls else "strait")
        res = []
        for general in ("1", "0"):
            os.environ["SZ_TW_GENERAL_CLIP"] = general
            w = fields.build_world(subzero_jl_amd.World(0), cfg)
            w.set_two_way(True, dt=cfg["dt"]); w.set_temps(0.3, -5.0)
            w.run(5, 0, cfg["dt"], coupling_dt=1)
            res.append(([a.copy() for a in w.ocean_stress()], w.get("u").copy(), w.get("cx").copy()))
        (sa, ua, xa), (sb, ub, xb) = res
        e = max(np.abs(a - b).max() / max(np.abs(a).max(), 1e-300) for a, b in zip(sa[:3], sb[:3]))
        eu = np.abs(ua - ub).max() / np.abs(ua).max()
        worst = max(worst, e, eu)
        print(f"seed {seed} walls {walls} conc {conc}: fields {e:.1e}  u {eu:.1e}  si_frac mean {sa[2].mean():.3f}", flush=True)
print("worst", worst)
assert worst < 1e-9
