"""single context: a resident run must not depend on whether the largest narrow variant is enqueued in every step (SZ_LEAN_NARROW=0) or left out until
an item needs it (a pause inside the step, default), nor on where the batches end -- fast fields between walls with topography under a sheared
ocean (items for the largest variant do occur), and fast periodic fields:   python tools/probe/lean_ab_fuzz.py [cases] [seed0]"""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np


def main():
    import subzero_jl_amd
    from subzero_jl_amd import fields
    from tests import test_tiles_gpu as T
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    ok = 0
    for c in range(cases):
        rnd = random.Random(seed0 + c)
        n = rnd.randrange(600, 2400); steps = rnd.randrange(20, 60); k = rnd.randrange(3, 12)
        shape = rnd.choice(["walls-topo", "walls-topo", "walls", "star"])
        cfg = T._field(n, seed0 + c, fast=True, shape=shape)
        os.environ.pop("SZ_LEAN_NARROW", None)
        a = fields.build_world(subzero_jl_amd.World(0), cfg)
        a.run(steps, 0, cfg["dt"], coupling_dt=1, stop_on_tags=False)
        retried = a.stats()["n_retry"]
        os.environ["SZ_LEAN_NARROW"] = "0"
        b = fields.build_world(subzero_jl_amd.World(0), cfg)
        os.environ.pop("SZ_LEAN_NARROW", None)
        b.run(steps, 0, cfg["dt"], coupling_dt=1, stop_on_tags=False)
        d = fields.build_world(subzero_jl_amd.World(0), cfg)
        for t0 in range(0, steps, k):
            d.run(min(k, steps - t0), t0, cfg["dt"], coupling_dt=1, stop_on_tags=False)
        bad = [f for f in T.FIELDS if not (np.array_equal(a.get(f)[:n], b.get(f)[:n]) and np.array_equal(a.get(f)[:n], d.get(f)[:n]))]
        ok += not bad
        print(f"case {c}: {shape} n {n} steps {steps} (batches of {k}); items handed to the largest variant: {retried}:", "identical" if not bad else f"DIFFER in {bad}", flush=True)
    print(f"{ok} / {cases} identical")


if __name__ == "__main__":
    main()
