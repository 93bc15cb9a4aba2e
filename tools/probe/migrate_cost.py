"""Cost of one re-tile (sz_tile_migrate) with the movers packed on the device against the host-staged path (SZ_MIGRATE_HOST=1):
`world` ranks sharing the GPU over the host transport (gloo), fast floes, a re-tile every 20 steps.
usage: python tools/probe/migrate_cost.py [n_floes] [world]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    import torch.multiprocessing as mp
    from tests import test_tiles_gpu as T
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    for host in ("0", "1"):
        os.environ["SZ_MIGRATE_HOST"] = host
        os.environ["SZ_PROBE_ANY_PATH"] = "1"
        ctx = mp.get_context("spawn")
        q = ctx.Queue(); port = T._free_port()
        procs = [ctx.Process(target=T._run_worker_migrate, args=(r, world, port, n, 78, 60, 20, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = T._collect(q, world)
        for p in procs:
            p.join(120)
        for rank, gidx, out, mv, cost in sorted(res, key=lambda r: r[0]):
            print(f"SZ_MIGRATE_HOST={host} n={n} rank {rank}/{world}: {len(gidx)} owned, {mv} given away, re-tile cost {['%.2f ms' % (1e3 * c) for c in cost]}", flush=True)


if __name__ == "__main__":
    main()
