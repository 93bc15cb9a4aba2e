"""does a resident run depend on where its batches end?  One sz_step batch of T steps against T batches of one step (and batches of k), single context,
the fast periodic field of the tile tests:   python tools/probe/batch_split_diff.py <n> <seed> <T> [k]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np


def main():
    import subzero_jl_amd
    from subzero_jl_amd import fields
    from tests import test_tiles_gpu as T
    n, seed, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    cfg = T._field(n, seed, fast=True)
    a = fields.build_world(subzero_jl_amd.World(0), cfg)
    b = fields.build_world(subzero_jl_amd.World(0), cfg)
    first = None
    for t0 in range(0, steps, k):
        m = min(k, steps - t0)
        b.run(m, t0, cfg["dt"], coupling_dt=1, stop_on_tags=False)
        a.run(m, t0, cfg["dt"], coupling_dt=1, stop_on_tags=False) if False else None
    a.run(steps, 0, cfg["dt"], coupling_dt=1, stop_on_tags=False)
    bad = {}
    for f in T.FIELDS:
        d = np.nonzero(a.get(f)[:n] != b.get(f)[:n])[0]
        if len(d):
            bad[f] = (len(d), d[:5].tolist(), float(np.max(np.abs(a.get(f)[:n] - b.get(f)[:n]))))
    print(f"n {n} seed {seed}: one batch of {steps} steps against batches of {k}:", "identical" if not bad else bad)


if __name__ == "__main__":
    main()
