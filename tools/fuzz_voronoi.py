"""First collision call on many Voronoi fields (touching cells: every contact degenerate) -- HIP path against the oracle: ghosts,
pair lists (bit-exact) and interaction rows (1e-10), periodic and walled.  usage: python tools/fuzz_voronoi.py [nseeds] [n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_parity as T
import parity
from subzero_jl_amd import fields

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
t0 = time.time(); bad = 0
for seed in range(1000, 1000 + nseeds):
    walls = seed % 2 == 1
    conc = [0.6, 0.8, 1.0][seed % 3]
    cfg = fields.make_config(n_floes=n, seed=seed, spacing=1.0e4, shape="voronoi", ocean="shear", concentration=conc, walls=walls)
    # jiggle the cells a little so that the contacts are slivers, not only exact touches (every third seed stays exact)
    if seed % 3:
        rng = np.random.default_rng(seed)
        dx = rng.uniform(-30, 30, n); dy = rng.uniform(-30, 30, n)
        for i in range(n):
            o0, o1 = cfg["vert_off"][i], cfg["vert_off"][i + 1]
            cfg["vx"][o0:o1] += dx[i]; cfg["vy"][o0:o1] += dy[i]
        from subzero_jl_amd import floe as floe_mod
        cfg["derived"] = floe_mod.derive(cfg["vert_off"], cfg["vx"], cfg["vy"], cfg["height"])
    try:
        hw, ow = T._pair(cfg)
        for w in (hw, ow):
            w.add_ghosts(); w.timestep_collisions(n, cfg["dt"])
        assert hw.M == ow.M and hw.ghosts() == ow.ghosts(), "ghosts"
        npairs = parity.compare_pairs(hw, ow)
        parity.compare_interactions(hw, ow, rtol=1e-10)
        print(f"ok   seed {seed} walls {int(walls)} conc {conc} pairs {npairs} rows {len(ow.interactions()[1])}", flush=True)
    except Exception as e:        # noqa: BLE001
        bad += 1
        print(f"FAIL seed {seed} walls {int(walls)} conc {conc}: {str(e)[:300]}", flush=True)
print(f"{nseeds - bad}/{nseeds} fields agree ({time.time() - t0:.0f} s)")
sys.exit(1 if bad else 0)
