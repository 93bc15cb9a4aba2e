import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import subzero_jl_amd
from subzero_jl_amd import fields
import test_tiles_gpu as T
for cand in range(5500, 6200, 50):
    hw = fields.build_world(subzero_jl_amd.World(0), T._pause_and_tag_cfg(float(cand)))
    pause = fuse = None
    for k in range(14):
        d = hw.run(1, k, 10, coupling_dt=10, coupling_on=False, stop_on_tags=True)
        if pause is None and hw.stats()["n_retry"] >= 1: pause = k + 1
        if fuse is None and np.any(hw.get("status")[2:] != 1): fuse = k + 1
    print(cand, pause, fuse, hw.get("status"))
