"""cost of the process-mode boundary at the bench size: sz_upload_floes (all columns), one timestep_collisions!,
download of the interactions -- what a shim that keeps the state on the host pays per replaced call.
usage: python tools/upload_bench.py [n_floes]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subzero_jl_amd
from subzero_jl_amd import fields

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfg = fields.make_config(n_floes=n, seed=12345)
w = fields.build_world(subzero_jl_amd.World(0), cfg)
w.add_ghosts(); w.timestep_collisions(n, cfg["dt"]); w.remove_ghosts()
for name, fn in (("upload (all columns)", lambda: (setattr(w, "_dirty", True), w._push())),
                 ("add_ghosts + timestep_collisions + remove_ghosts", lambda: (w.add_ghosts(), w.timestep_collisions(n, cfg["dt"]), w.remove_ghosts())),
                 ("download interactions", lambda: (setattr(w, "_host_stale", True), w.interactions())),
                 ("download all columns", lambda: (setattr(w, "_host_stale", True), w._pull()))):
    fn()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    print(f"{name:52s} {(time.perf_counter() - t0) / 10 * 1e3:8.2f} ms", flush=True)
