import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import subzero_jl_amd
from subzero_jl_amd import fields
pre = len(sys.argv) > 1 and sys.argv[1] == "pause"
if pre:
    import test_hip_parity as T
    hw = T._retry_scenario(T.mk())
    print("pause scenario:", hw.run(8, 0, 10, coupling_dt=10, coupling_on=False, stop_on_tags=False), hw.stats()["n_retry"], flush=True)
    del hw
cfg = fields.make_config(n_floes=1200, seed=21, concentration=0.8)
hw = fields.build_world(subzero_jl_amd.World(0), cfg)
rng = np.random.default_rng(3)
hw.set("u", rng.uniform(-40.0, 40.0, cfg["n_floes"])); hw.set("v", rng.uniform(-40.0, 40.0, cfg["n_floes"]))
t = 0
for k in (1, 7, 3, 1, 12):
    print("batch of", k, "from", t, flush=True)
    r = hw.run(k, t, cfg["dt"], coupling_dt=1, stop_on_tags=False)
    t += k
    st = hw.stats()
    print("  ran", r, "M", st["M"], "ghosts", st["n_ghosts"], "retry", st["n_retry"], "fuse", st["n_status_fuse"], "remove", st["n_status_remove"], "tracefail", st["n_trace_fail"], "pairs", st["n_pairs"], "rows", st["n_inter_rows"], "mismatch", hw.crec_mismatches(), flush=True)
print("done", flush=True)
