"""Tiled run with `world` ranks sharing the one GPU of the box (gloo, host staging) against the single-context run:
bit-equal columns of every owned floe.  The parent touches the GPU only after the workers have finished, so up to 6
ranks stay within the box's process limit.  usage: python tools/tiles_ranks.py [world] [n_floes] [steps] [torch|library-host]
(library-host: the exchange inside the library, sz_tile_run, with the transfers over gloo -- sz_comm_init_host)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_tiles_gpu as T


def main():
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    backend = sys.argv[4] if len(sys.argv) > 4 else "torch"
    assert world <= 6, "at most 6 processes may use the GPU together"
    seed = 37
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = T._free_port()
    procs = [ctx.Process(target=T._run_worker, args=(r, world, port, n, seed, steps, q, False, False, backend)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = T._collect(q, world)
        for p in procs:
            p.join(120)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    import subzero_jl_amd
    from subzero_jl_amd import fields, tiles
    cfg = fields.make_config(n_floes=n, seed=seed)
    hw = fields.build_world(subzero_jl_amd.World(0), cfg)
    hw.run(steps, 0, cfg["dt"], coupling_dt=1, stop_on_tags=False)
    seen = np.zeros(n, bool)
    for rank, gidx, out, nhalo, vx in res:
        seen[gidx] = True
        for f in T.FIELDS:
            assert np.array_equal(out[f], hw.get(f)[gidx]), (rank, f)
        print(f"rank {rank}: {len(gidx)} floes, {nhalo} halo floes, bit-equal", flush=True)
    assert seen.all()
    print(f"{world} ranks ({'x'.join(map(str, tiles.tile_grid(world)))} tiles) == single context, {n} floes, {steps} steps, exchange: {backend}")


if __name__ == "__main__":
    main()
