"""first step at which a tiled run with re-tiling (a case of tools/fuzz_tiles.py) leaves the single context's trajectory, and the floes that do:
    python tools/probe/tiles_first_diff.py <seed> [mixed | walls] [no-migrate | batches-only]"""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np


def run(T_, world, n, seed, steps, every, shape="star", fast=True, stop=False):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = T_._free_port()
    procs = [ctx.Process(target=T_._run_worker_migrate, args=(r, world, port, n, seed, steps, every, q, shape, fast, stop)) for r in range(world)]
    for p in procs:
        p.start()
    res = T_._collect(q, world)
    for p in procs:
        p.join(60)
    return res


def main():
    import subzero_jl_amd
    from subzero_jl_amd import fields
    from tests import test_tiles_gpu as T
    os.environ["SZ_PROBE_ANY_PATH"] = "1"
    seed = int(sys.argv[1])
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import fuzz_tiles
    opts = sys.argv[2:]
    world, n, steps, every, kind, shape, fast, stop = fuzz_tiles.case_params(seed, "mixed" in opts, "walls" in opts)
    if "no-migrate" in opts:
        every = 10 ** 6               # one batch
    if "batches-only" in opts:
        os.environ["SZ_PROBE_SKIP_MIGRATE"] = "1"          # the same batches, no re-tile between them
    cfg = T._field(n, seed, fast=fast, shape=shape)

    def diffs(Tn):
        res = run(T, world, n, seed, Tn, every, shape, fast, stop)
        hw = fields.build_world(subzero_jl_amd.World(0), cfg)
        hw.run(Tn, 0, cfg["dt"], coupling_dt=1, stop_on_tags=stop)
        bad = []
        tagged = np.nonzero(hw.get("status")[:n] != 1)[0]
        if len(tagged):
            print(f"   after {Tn} steps the single context holds tagged floes {tagged.tolist()} (status {hw.get('status')[tagged].tolist()}), steps run there / here:",
                  [r[2]["_ran"] for r in res])
        for rank, gidx, out, mv, cost in res:
            if out["_ran"] != Tn:
                bad.append((-1, rank, "steps run", float(out["_ran"]), float(Tn)))
            for f in T.FIELDS:
                ref = hw.get(f)[gidx]
                for k in np.nonzero(out[f] != ref)[0]:
                    bad.append((int(gidx[k]), rank, f, float(out[f][k]), float(ref[k])))
            off, x, y = out["_rings"]; hoff, hx, hy = hw.rings()
            for k, g in enumerate(gidx):
                for nm, a, b in (("ring x", x[off[k]:off[k + 1]], hx[hoff[g]:hoff[g + 1]]), ("ring y", y[off[k]:off[k + 1]], hy[hoff[g]:hoff[g + 1]])):
                    for j in np.nonzero(a != b)[0]:
                        bad.append((int(g), rank, f"{nm}[{j}]", float(a[j]), float(b[j])))
        return bad, hw
    lo, hi = 0, steps            # state equal after lo steps, differs after hi
    bad, hw = diffs(hi)
    if not bad:
        print("no difference after", steps, "steps"); return
    while hi - lo > 1:
        mid = (lo + hi) // 2
        b, _ = diffs(mid)
        if b: hi = mid
        else: lo = mid
    bad, hw = diffs(hi)
    L = cfg["L"]
    print(f"seed {seed} [{kind}]: world {world} n {n} steps {steps} re-tile every {every}: first difference after step {hi}")
    floes = sorted(set(b[0] for b in bad))
    for g in floes[:8]:
        cx, cy, r = hw.get("cx")[g], hw.get("cy")[g], hw.get("rmax")[g]
        print(f"  floe {g}: cx/L {cx / L:.4f} cy/L {cy / L:.4f} rmax/L {r / L:.4f} overarea {hw.get('overarea')[g]:.3g}; fields:",
              [(b[2], b[3], b[4]) for b in bad if b[0] == g][:6])


if __name__ == "__main__":
    main()
