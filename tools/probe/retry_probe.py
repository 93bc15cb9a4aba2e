"""the retry scenario of tests/test_hip_parity.py (an item for the largest narrow variant) step by step against the oracle: totals and stress columns"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_hip_parity as T
os.environ["SZ_LEAN_NARROW"] = sys.argv[1] if len(sys.argv) > 1 else "0"
hw = T._retry_scenario(T.mk()); ow = T._retry_scenario(T.omk())
for k in range(10):
    hw.run(1, k, 10, coupling_dt=10, coupling_on=False, stop_on_tags=False)
    ow.timestep_sim(k, 10, coupling_dt=10, coupling_on=False)
    ho, hr = hw.interactions(); oo, orr = ow.interactions()
    print("step", k, "rows", len(hr), len(orr), "retries", hw.stats()["n_retry"])
    for f in ("coll_fx", "coll_fy", "coll_trq", "si11", "si12", "si22", "sa22", "overarea"):
        a, b = hw.get(f), ow.get(f)
        print(f"   {f:9s} hip {a[:2]} oracle {b[:2]}")
