"""Where do the HIP path and the oracle part on a Voronoi field?  Steps both one step at a time and prints the largest relative
difference of the state columns, the pair-list equality and the worst interaction row.  usage: python tools/voronoi_diverge.py [n seed walls conc steps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_parity as T
import parity
from subzero_jl_amd import fields

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4
walls = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
conc = float(sys.argv[4]) if len(sys.argv) > 4 else 0.85
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 30
cfg = fields.make_config(n_floes=n, seed=seed, spacing=1.0e4, shape="voronoi", ocean="shear", concentration=conc, walls=walls)
hw, ow = T._pair(cfg)
for t in range(steps):
    hw.run(1, t, cfg["dt"], coupling_dt=5, stop_on_tags=False)
    ow.timestep_sim(t, cfg["dt"], coupling_dt=5)
    d = {f: parity.relerr(hw.get(f), ow.get(f)) for f in ("cx", "u", "v", "xi", "alpha", "coll_fx", "coll_trq", "overarea")}
    hp, op = hw.pairs(), ow.pairs()
    same = len(hp[0]) == len(op[0]) and np.array_equal(hp[0], op[0]) and np.array_equal(hp[1], op[1])
    ho, hr = hw.interactions(); oo, orr = ow.interactions()
    nrow = (len(hr), len(orr))
    worst = ""
    if len(hr) == len(orr) and len(hr):
        dd = np.abs(hr - orr); k = np.unravel_index(np.argmax(dd / (np.abs(orr) + 1e-3)), dd.shape)
        worst = f"row {k} hip {hr[k]!r} oracle {orr[k]!r}"
    print(t, "pairs equal" if same else "PAIRS DIFFER", "rows", nrow, {k: f"{v:.1e}" for k, v in d.items()}, worst, flush=True)
