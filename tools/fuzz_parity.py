"""Wider parity sweep than the test suite: random fields at several concentrations / boundary kinds / seeds,
a few resident steps each, HIP path against the CPU oracle (pairs bit-exact, state within 1e-9)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import parity
import subzero_jl_amd
from subzero_jl_amd import fields
from oracle import orc

cases = []
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for conc in (0.6, 0.8, 0.98):
        for walls in (False, True):
            cases.append(dict(n_floes=700, seed=100 + seed, concentration=conc, walls=walls, topography=walls,
                              ocean=("strait" if walls else "converge_diverge")))
bad = 0
t0 = time.time()
for k, kw in enumerate(cases):
    cfg = fields.make_config(**kw)
    hw = fields.build_world(subzero_jl_amd.World(0), cfg); ow = fields.build_world(orc.World(), cfg); ow.set_threads(8)
    steps = 6
    try:
        hw.run(steps, 0, cfg["dt"], coupling_dt=2)
        for t in range(steps):
            ow.timestep_sim(t, cfg["dt"], coupling_dt=2)
        parity.compare_worlds(hw, ow, rtol=1e-9)
        npairs = parity.compare_pairs(hw, ow)
        st = hw.stats()
        print(f"ok   {kw}  pairs {npairs} rows {st['n_inter_rows']} retry {st['n_retry']} tracefail {st['n_trace_fail']}", flush=True)
    except AssertionError as e:
        bad += 1
        print(f"FAIL {kw}: {str(e)[:300]}", flush=True)
print(f"{len(cases) - bad}/{len(cases)} cases agree ({time.time() - t0:.0f} s)")


# ---- second sweep: rough shapes -- 3 to 40 vertices, deep radial variation (concave, spiky), heavy overlaps:
# several contact regions per pair, many crossings, rings above the small kernels' capacity, working-set retries
from rough_shapes import rough_world


bad2 = 0; n2 = 0
for seed in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    for periodic in (True, False):
        hw = rough_world(lambda: subzero_jl_amd.World(0), 500 + seed, 12, periodic)
        ow = rough_world(orc.World, 500 + seed, 12, periodic)
        n2 += 1
        try:
            for w in (hw, ow):
                n = w.M
                w.add_ghosts(); w.timestep_collisions(n, 10)
            npairs = parity.compare_pairs(hw, ow)
            parity.compare_interactions(hw, ow, 1e-9)
            parity.compare_worlds(hw, ow, rtol=1e-9, fields=["coll_fx", "coll_fy", "coll_trq", "overarea"])
            st = hw.stats(); rows = ow.interactions()[1]
            print(f"ok   rough seed {seed} periodic {periodic}: pairs {npairs} rows {len(rows)} retry {st['n_retry']} "
                  f"tracefail {st['n_trace_fail']} oracle tracefail {orc.lib().orc_trace_failures()}", flush=True)
        except AssertionError as e:
            bad2 += 1
            print(f"FAIL rough seed {seed} periodic {periodic}: {str(e)[:300]}", flush=True)
        except subzero_jl_amd.capi.SzError as e:          # a capacity the engine reports loudly (never a silent drop)
            print(f"cap  rough seed {seed} periodic {periodic}: {str(e)[:120]}", flush=True)
print(f"{n2 - bad2}/{n2} rough cases agree")
sys.exit(1 if (bad or bad2) else 0)
