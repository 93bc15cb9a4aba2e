"""Soak run: many timesteps of the bench workload, stats every few hundred steps (capacity, retries, trace failures,
finite state, step time as the contact network evolves)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import subzero_jl_amd
from subzero_jl_amd import fields

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
total = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
kw = dict(walls=True, topography=True, ocean="strait") if (len(sys.argv) > 3 and sys.argv[3] == "walls") else {}
cfg = fields.make_config(n_floes=n, seed=12345, **kw)
w = fields.build_world(subzero_jl_amd.World(0), cfg)
t = 0
while t < total:
    k = min(500, total - t)
    t0 = time.perf_counter(); w.run(k, t, cfg["dt"], coupling_dt=1); el = time.perf_counter() - t0
    t += k
    st = w.stats()
    u = w.get("u"); cx = w.get("cx"); status = w.ids()[2]
    print(f"step {t}: {1e3 * el / k:.4f} ms/step  pairs {st['n_pairs']} run {st['n_pairs_clipped']} rows {st['n_inter_rows']} ghosts {st['n_ghosts']} "
          f"retry {st['n_retry']} tracefail {st['n_trace_fail']} finite {bool(np.isfinite(u).all() and np.isfinite(cx).all())} "
          f"max|u| {np.abs(u).max():.3f} status!=active {(status != 1).sum()} warn {w.warn_counts()}", flush=True)
