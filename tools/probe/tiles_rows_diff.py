"""interaction rows of one floe after T steps: a tiled run with re-tiling (a case of tools/fuzz_tiles.py) against the single context
    python tools/probe/tiles_rows_diff.py <seed> <T> <global floe index>"""
import os, sys, random, datetime
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np


def worker(rank, world, port, n, seed, steps, every, target, q):
    import torch.distributed as dist
    from subzero_jl_amd import tiles
    from tests import test_tiles_gpu as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        cfg = T._field(n, seed, fast=True)
        tw = tiles.TiledWorld(cfg, rank, world, 0, dist, host_staging=True, backend="library-host", rebox_every=5)
        hist = []
        for t0 in range(0, steps, every):
            hist.append((t0, target in set(tw.gidx.tolist())))
            tw.run(min(every, steps - t0), t0, cfg["dt"], coupling_dt=1)
            if t0 + every < steps and not os.environ.get("SZ_PROBE_SKIP_MIGRATE"):
                tw.migrate()
        rows = None; ghost = None
        g = tw.gidx.tolist()
        if target in g:
            rows = tw.world.inter(g.index(target)).copy()
            key = int(os.environ.get("SZ_PROBE_KEY", "0"))
            if key:
                last = steps - (steps - 1) // every * every          # steps of the last batch: the allocator alternates from the start of a batch
                ghost = tw.world.find_key((last - 1) & 1, key)
                par = (key & ((1 << 40) - 1)) >> 2
                ghost = dict(ghost or {}, pairs=tw.world.pairs_of_ids((last - 1) & 1, target + 1, par + 1),
                             instances={kk: (lambda r: None if r is None else (r["row"], r["cx"], r["cy"], r["parent"]))(tw.world.find_key((last - 1) & 1, kk)) for kk in
                                        [(1 << 40) + 4 * target, (2 << 40) + 4 * target, (2 << 40) + 4 * target + 1, (1 << 40) + 4 * par, (2 << 40) + 4 * par, (2 << 40) + 4 * par + 1, par]})
        q.put((rank, hist, rows, {f: tw.owned(f)[g.index(target)] for f in ("cx", "cy", "overarea", "coll_fx")} if target in g else None, ghost))
    finally:
        dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    import subzero_jl_amd
    from subzero_jl_amd import fields
    from tests import test_tiles_gpu as T
    os.environ["SZ_PROBE_ANY_PATH"] = "1"
    seed, Tn, target = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    rnd = random.Random(seed)
    world = rnd.choice([2, 2, 4])
    n = rnd.randrange(500, 1600) if world == 2 else rnd.randrange(1000, 2400)
    every = rnd.randrange(4, 16)
    cfg = T._field(n, seed, fast=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = T._free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, n, seed, Tn, every, target, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    hw = fields.build_world(subzero_jl_amd.World(0), cfg)
    hw.run(Tn, 0, cfg["dt"], coupling_dt=1, stop_on_tags=False)
    ref = hw.inter(target)
    np.set_printoptions(precision=17, linewidth=250)
    print("single context rows of floe", target, "after", Tn, "steps:\n", ref)
    key = int(os.environ.get("SZ_PROBE_KEY", "0"))
    gref = hw.find_key((Tn - 1) & 1, key) if key else None
    if key:
        par = (key & ((1 << 40) - 1)) >> 2
        print("single context: ghost with key", key, "(parent", par, "):", gref)
        print("single context: pair items between the instances of the two ids:", hw.pairs_of_ids((Tn - 1) & 1, target + 1, par + 1))
        print("single context: instances present:", {kk: (lambda r: None if r is None else (r["row"], r["cx"], r["cy"], r["parent"]))(hw.find_key((Tn - 1) & 1, kk)) for kk in
              [(1 << 40) + 4 * target, (2 << 40) + 4 * target, (2 << 40) + 4 * target + 1, (1 << 40) + 4 * par, (2 << 40) + 4 * par, (2 << 40) + 4 * par + 1]})
        print("single context: the parent now: cx cy", hw.get("cx")[par], hw.get("cy")[par], "ring", hw.ring(par).T)
    for rank, hist, rows, vals, ghost in sorted(res, key=lambda r: r[0]):
        if ghost is not None:
            print("rank", rank, ": pair items:", ghost.get("pairs"), "instances present:", ghost.get("instances"))
            if gref is not None and "cx" in ghost:
                print("   equal to the single context's: cx", ghost["cx"] == gref["cx"], "cy", ghost["cy"] == gref["cy"], "x", np.array_equal(ghost["x"], gref["x"]),
                      "y", np.array_equal(ghost["y"], gref["y"]), "u v xi", ghost["u"] == gref["u"], ghost["v"] == gref["v"], ghost["xi"] == gref["xi"],
                      "box", np.array_equal(ghost["box"], gref["box"]))
                print("   dx", ghost["x"] - gref["x"], "dy", ghost["y"] - gref["y"], "dcx", ghost["cx"] - gref["cx"], "dcy", ghost["cy"] - gref["cy"])
        print("rank", rank, "owned the floe at the start of the batches", hist)
        if rows is not None:
            print("tiled rows:\n", rows)
            print("values", vals, "single:", {f: hw.get(f)[target] for f in vals})
            if rows.shape == ref.shape:
                print("row-wise equal (columns 1..6):", [bool(np.array_equal(rows[k, 1:], ref[k, 1:])) for k in range(len(ref))])
                a = sorted(map(tuple, rows[:, 1:].tolist())); b = sorted(map(tuple, ref[:, 1:].tolist()))
                print("equal as multisets:", a == b)


if __name__ == "__main__":
    main()
