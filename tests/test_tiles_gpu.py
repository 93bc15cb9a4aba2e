"""Tiled (multi-rank) engine on the real GPU: 2 ranks share the one MI355X of the test box and trade
their halos through gloo (host staging); results must equal the single-context run."""
import datetime
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ["cx", "cy", "u", "v", "xi", "alpha", "coll_fx", "coll_fy", "coll_trq", "fxOA", "fyOA", "trqOA", "overarea",
          "sa11", "sa22", "e11", "e22"]


def _guard(fn):
    """run a worker body; an exception goes to the parent through the queue instead of leaving it waiting"""
    def run(rank, world, port, *args):
        q = next(a for a in args if hasattr(a, "put"))
        try:
            fn(rank, world, port, *args)
        except BaseException as e:          # noqa: BLE001
            import traceback
            q.put(("error", rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:]))
            raise
    return run


def _collect(q, world):
    res = []
    for _ in range(world):
        r = q.get(timeout=180)
        assert r[0] != "error", f"rank {r[1]} failed:\n{r[2]}"
        res.append(r)
    return res


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _field(n, seed, fast=False, shape="star"):
    from subzero_jl_amd import fields
    from subzero_jl_amd import floe as floe_mod
    kw = {}
    if shape == "voronoi":
        kw = {"spacing": 1.0e4, "ocean": "shear"}
    elif shape in ("walls", "walls-topo"):          # collision walls instead of periodic ones (configs[3]), with or without the island and the coast wedges
        kw = {"walls": True, "topography": shape == "walls-topo", "ocean": "shear"}; shape = "star"
    cfg = fields.make_config(n_floes=n, seed=seed, shape=shape, **kw)
    if fast:
        # fast floes on a field shifted so that parents straddle the walls: some start outside the domain, others leave it during the
        # run -- they swap with their ghosts (collisions.jl:942-950), and the forcings of a tiled step are evaluated BEFORE its ghost pass
        cfg["u"] = np.abs(cfg["u"]) * 60.0 + 2.0; cfg["v"] = cfg["v"] * 60.0 + 1.0
        cfg["vx"] = cfg["vx"] + 9800.0; cfg["vy"] = cfg["vy"] + 9900.0
        cfg["derived"] = floe_mod.derive(cfg["vert_off"], cfg["vx"], cfg["vy"], cfg["height"])
    return cfg


def _worker(rank, world, port, n, seed, steps, q, repartition=False, fast=False, backend="torch", shape="star"):
    import torch
    import torch.distributed as dist
    from subzero_jl_amd import fields, tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        cfg = _field(n, seed, fast, shape)
        tw = tiles.TiledWorld(cfg, rank, world, 0, dist, host_staging=True, backend=backend, rebox_every=3 if backend != "torch" else 50)
        if repartition:
            # half way through, hand every floe to the other kind of tiling (split along y instead of x): most
            # floes change owner with their complete state, and the run must go on as if nothing had happened
            half = steps // 2
            tw.run(half, 0, cfg["dt"], coupling_dt=1)
            L = cfg["L"]
            other = lambda cx, cy: (cy > 0.5 * L).astype(int)
            if repartition == "migrate-band":          # a band either side of the tile edge changes hands: the tiles keep their size
                other = lambda cx, cy: ((cx > 0.5 * L) ^ (np.abs(cx - 0.5 * L) < 0.06 * L)).astype(int)
            elif repartition == "migrate-lopsided":    # one tile grows by a sixth: more than its context was carved to take in
                other = lambda cx, cy: (cx > 0.40 * L).astype(int)
            if repartition == "migrate-host":
                os.environ["SZ_MIGRATE_HOST"] = "1"
            moved = tw.migrate(owner_fn=other) if str(repartition).startswith("migrate") else tw.repartition(owner_fn=other)
            assert (moved > 0 or repartition == "migrate-lopsided") and len(tw.gidx) > 0          # (lopsided: one rank only receives)
            if str(repartition).startswith("migrate"):
                # packed on the device (1) unless a tile outgrows its context or the switch asks for the host-staged path (2): the same on every rank
                want = {"migrate-band": (1,), "migrate-host": (2,), "migrate-lopsided": (2,)}.get(repartition, (1, 2))
                assert tw.migrate_path in want, (repartition, tw.migrate_path)
                paths = [None] * world
                dist.all_gather_object(paths, tw.migrate_path)
                assert len(set(paths)) == 1, paths
            tw.run(steps - half, half, cfg["dt"], coupling_dt=1)
        else:
            tw.run(steps, 0, cfg["dt"], coupling_dt=1)
        out = {f: tw.owned(f) for f in FIELDS}
        off, x, y = tw.world.rings()
        q.put((rank, tw.gidx, out, tw.n_halo_last, x[:off[len(tw.gidx)]].copy()))
    finally:
        dist.destroy_process_group()


def _run_worker(*a):
    _guard(_worker)(*a)


def _run_worker_two_way(*a):
    _guard(_worker_two_way)(*a)


def _cfg2():
    from subzero_jl_amd import fields
    return fields.make_config(n_floes=100000, seed=12346, ocean="converge_diverge")          # bench.py's configs[2] field


def _worker_cfg2(rank, world, port, steps, q):
    import torch.distributed as dist
    from subzero_jl_amd import tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    try:
        cfg = _cfg2()
        tw = tiles.TiledWorld(cfg, rank, world, 0, dist, host_staging=True, backend="library-host", rebox_every=50)
        tw.run(steps, 0, cfg["dt"], coupling_dt=1)
        q.put((rank, tw.gidx, {f: tw.owned(f) for f in FIELDS}, tw.n_halo_last, None))
    finally:
        dist.destroy_process_group()


def _run_worker_cfg2(*a):
    _guard(_worker_cfg2)(*a)


def test_configs2_100k_in_four_tiles_equals_the_single_context():
    """BASELINE configs[2] at the size the multi-GPU metric is quoted on: 100 000 floes, converge / diverge ocean, cut into 2 x 2 tiles (four
    ranks sharing the box's GPU, the library's exchange over the host channel), three coupled timesteps -- every owned column of every
    rank bit-equal to the single context's run of the same field."""
    import torch.multiprocessing as mp
    import subzero_jl_amd
    from subzero_jl_amd import fields
    world, steps = 4, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_run_worker_cfg2, args=(r, world, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = []
        for _ in range(world):
            r = q.get(timeout=600)
            assert r[0] != "error", f"rank {r[1]} failed:\n{r[2]}"
            res.append(r)
        for p in procs:
            p.join(120)
        assert all(p.exitcode == 0 for p in procs)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    cfg = _cfg2()
    hw = fields.build_world(subzero_jl_amd.World(0), cfg)
    hw.run(steps, 0, cfg["dt"], coupling_dt=1, stop_on_tags=False)
    seen = np.zeros(cfg["n_floes"], bool)
    for rank, gidx, out, nhalo, _ in res:
        assert nhalo > 0 and 15000 < len(gidx) < 35000
        seen[gidx] = True
        for f in FIELDS:
            ref = hw.get(f)[gidx]
            assert np.array_equal(out[f], ref), (rank, f, np.max(np.abs(out[f] - ref)))
    assert seen.all()


@pytest.mark.parametrize("world,n,seed,steps,repartition,fast,backend", [
    (2, 600, 31, 4, False, False, "torch"), (2, 600, 33, 6, True, False, "torch"), (4, 1000, 35, 4, False, False, "torch"),
    (2, 500, 77, 30, False, True, "torch"),
    (2, 600, 31, 8, False, False, "library-host"), (2, 600, 33, 8, True, False, "library-host"), (2, 600, 33, 8, "migrate", False, "library-host"),
    (2, 600, 33, 8, "migrate-band", False, "library-host"), (2, 600, 33, 8, "migrate-host", False, "library-host"), (2, 600, 33, 8, "migrate-lopsided", False, "library-host"),
    (4, 1000, 35, 8, False, False, "library-host"),
    (2, 500, 77, 30, False, True, "library-host")])
def test_ranks_equal_single(world, n, seed, steps, repartition, fast, backend):
    """2 ranks (two tiles side by side) and 4 ranks (2 x 2 tiles: corner halos, both periodic directions across
    tile boundaries) against the single-context run: bit-equal columns for every owned floe; `fast`: parents cross the
    periodic walls and swap with their ghosts during the run.  backend "torch": the exchange is one torch.distributed
    collective per step on buffers the library packs; "library-host": the exchange INSIDE the library (sz_tile_run: box and
    count-matrix gathers, per-pair regions with real counts, neighbours only, drift reference, several gather intervals) with
    the ranks' transfers made by the host channel (sz_comm_init_host over gloo) -- every line of the multi-rank path except the
    RCCL calls themselves, which ranks sharing one GPU cannot make."""
    import torch.multiprocessing as mp
    import subzero_jl_amd
    from subzero_jl_amd import fields
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_worker, args=(r, world, port, n, seed, steps, q, repartition, fast, backend)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, world)
        for p in procs:
            p.join(60)
        assert all(p.exitcode == 0 for p in procs)
    finally:
        for p in procs:                      # a rank left waiting in a collective by a failed peer
            if p.is_alive():
                p.terminate()
    cfg = _field(n, seed, fast)
    hw = fields.build_world(subzero_jl_amd.World(0), cfg)
    hw.run(steps, 0, cfg["dt"], coupling_dt=1, stop_on_tags=False)
    if fast:
        wrapped = np.abs(hw.get("cx") - cfg["derived"]["cx"]) > 0.5 * cfg["L"]
        assert wrapped.sum() >= 3                  # parents went through a wall and came back on the other side
    seen = np.zeros(n, bool)
    for rank, gidx, out, nhalo, vx in res:
        assert nhalo > 0
        seen[gidx] = True
        for f in FIELDS:
            ref = hw.get(f)[gidx]
            assert np.array_equal(out[f], ref), (rank, f, np.max(np.abs(out[f] - ref)))
    assert seen.all()


@pytest.mark.parametrize("world,backend,repartition", [(2, "torch", False), (4, "library-host", False), (2, "library-host", "migrate-band")])
def test_voronoi_field_ranks_equal_single(world, backend, repartition):
    """The reference's own kind of field -- touching Voronoi cells with a size spectrum (the larger neighbour capacity, the chunked
    candidate pool, halo floes the upload-time count has not seen) -- in tiles: bit-equal to the single context.  "migrate-band": half way
    through, the cells either side of the tile edge change owner through sz_tile_migrate's device path (records of many ring and
    sub-floe point counts, rows of the larger capacities)."""
    import torch.multiprocessing as mp
    import subzero_jl_amd
    from subzero_jl_amd import fields
    n, seed, steps = (900 if world == 2 else 3600), 93, 10          # (a tile holds at most as many halo floes as owned ones: 4 tiles need the larger field)
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_run_worker, args=(r, world, port, n, seed, steps, q, repartition, False, backend, "voronoi")) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, world)
        for p in procs:
            p.join(60)
        assert all(p.exitcode == 0 for p in procs)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    cfg = _field(n, seed, False, "voronoi")
    hw = fields.build_world(subzero_jl_amd.World(0), cfg)
    hw.run(steps, 0, cfg["dt"], coupling_dt=1, stop_on_tags=False)
    seen = np.zeros(n, bool)
    for rank, gidx, out, nhalo, vx in res:
        assert nhalo > 0
        seen[gidx] = True
        for f in FIELDS:
            assert np.array_equal(out[f], hw.get(f)[gidx]), (rank, f, np.max(np.abs(out[f] - hw.get(f)[gidx])))
    assert seen.all() and np.count_nonzero(hw.get("overarea")) > n // 4


def _worker_two_way(rank, world, port, n, seed, steps, q, backend="torch"):
    import torch.distributed as dist
    from subzero_jl_amd import fields, tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        cfg = fields.make_config(n_floes=n, seed=seed, ocean="shear")
        tw = tiles.TiledWorld(cfg, rank, world, 0, dist, host_staging=True, backend=backend)
        tw.set_two_way(0.5, -8.0, cfg["dt"])
        tw.run(steps, 0, cfg["dt"], coupling_dt=2)
        tw.sync()
        L = cfg["L"]
        grid = tw.write_grid_data(np.linspace(0, L, 13), np.linspace(0, L, 10))      # GridOutputWriter averages across the tiles
        q.put((rank, tw.gidx, {f: tw.owned(f) for f in ("cx", "cy", "u", "v", "xi", "height")}, [a.copy() for a in tw.world.ocean_stress()], grid))
    finally:
        dist.destroy_process_group()


def _two_way_stop_cfg(n, seed):
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=n, seed=seed, concentration=0.8, ocean="shear")
    rng = np.random.default_rng(3)          # floes fast enough to run deep into one another within a few steps (a fuse: collisions.jl:366)
    cfg["u"] = rng.uniform(-40.0, 40.0, n); cfg["v"] = rng.uniform(-40.0, 40.0, n)
    return cfg


def _worker_two_way_stop(rank, world, port, n, seed, steps, q):
    import torch.distributed as dist
    from subzero_jl_amd import tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        cfg = _two_way_stop_cfg(n, seed)
        tw = tiles.TiledWorld(cfg, rank, world, 0, dist, host_staging=True, backend="library-host", rebox_every=3)
        tw.set_two_way(0.5, -8.0, cfg["dt"])
        done = tw.run(steps, 0, cfg["dt"], coupling_dt=2, stop_on_tags=True)
        tw.sync()
        q.put((rank, tw.gidx, {f: tw.owned(f) for f in ("cx", "cy", "u", "v", "xi", "height", "status")}, [a.copy() for a in tw.world.ocean_stress()], done))
    finally:
        dist.destroy_process_group()


def _run_worker_two_way_stop(*a):
    _guard(_worker_two_way_stop)(*a)


def test_two_way_coupling_across_tiles_ends_the_batch_on_a_tag():
    """simplify_floes! runs after every step in the reference (simulation.jl:205-214): a tiled batch with two-way coupling on ends, on every
    rank, after the step that tagged a floe -- it used to run to its end.  Fast floes that fuse a few steps in; two ranks against the
    single context: the same number of steps, the same tags, columns and ocean fields to the round-off of the cross-rank sums."""
    import torch.multiprocessing as mp
    import subzero_jl_amd
    from subzero_jl_amd import fields
    n, seed, steps, world = 500, 77, 30, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_run_worker_two_way_stop, args=(r, world, port, n, seed, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, world)
        for p in procs:
            p.join(60)
        assert all(p.exitcode == 0 for p in procs)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    cfg = _two_way_stop_cfg(n, seed)
    hw = fields.build_world(subzero_jl_amd.World(0), cfg)
    hw.set_two_way(True, dt=cfg["dt"]); hw.set_temps(0.5, -8.0)
    k = hw.run(steps, 0, cfg["dt"], coupling_dt=2, stop_on_tags=True)
    assert 1 <= k < steps and np.any(hw.get("status") != 1)          # a tag ended the batch in its middle
    ref = hw.ocean_stress()
    for rank, gidx, out, fields_, done in res:
        assert done == k, (rank, done, k)
        assert np.array_equal(out["status"], hw.get("status")[gidx]), rank
        for name, g, r in zip(("tau_x", "tau_y", "si_frac", "hflx"), fields_, ref):
            assert np.max(np.abs(g - r)) <= 1e-12 * max(np.max(np.abs(r)), 1e-300), (rank, name)
        for f, v in out.items():
            if f != "status":
                assert np.max(np.abs(v - hw.get(f)[gidx])) <= 1e-12 * np.max(np.abs(hw.get(f))), (rank, f)


@pytest.mark.parametrize("backend", ["torch", "library-host"])
def test_two_way_coupling_across_tiles(backend):
    """Two ranks, two-way coupling on: the per-cell sums of both ranks added up give the single-context ocean
    fields (to round-off: the cross-rank sum order differs), and the run -- which feeds on the heat-flux factor
    the coupling writes -- follows the single-context trajectory."""
    import torch.multiprocessing as mp
    import subzero_jl_amd
    from subzero_jl_amd import fields
    n, seed, steps, world = 500, 41, 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_run_worker_two_way, args=(r, world, port, n, seed, steps, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, world)
        for p in procs:
            p.join(60)
        assert all(p.exitcode == 0 for p in procs)
    finally:
        for p in procs:                      # a rank left waiting in a collective by a failed peer
            if p.is_alive():
                p.terminate()
    cfg = fields.make_config(n_floes=n, seed=seed, ocean="shear")
    hw = fields.build_world(subzero_jl_amd.World(0), cfg)
    hw.set_two_way(True, dt=cfg["dt"]); hw.set_temps(0.5, -8.0)
    hw.run(steps, 0, cfg["dt"], coupling_dt=2)
    ref = hw.ocean_stress()
    L = cfg["L"]
    gref = hw.write_grid_data(np.linspace(0, L, 13), np.linspace(0, L, 10))
    assert np.count_nonzero(gref[6]) > 60
    for rank, gidx, out, fields_, grid in res:
        # every rank holds the full grid output; the cross-rank sums differ from the serial ones by round-off only
        for k in range(len(gref)):
            scale = max(np.abs(gref[j]).max() for j in (range(9, 13) if 9 <= k <= 13 else range(14, 18) if k >= 14 else (k,)))
            assert np.abs(grid[k] - gref[k]).max() <= 1e-11 * scale, (rank, k)
        for name, g, r in zip(("tau_x", "tau_y", "si_frac", "hflx"), fields_, ref):
            assert np.max(np.abs(g - r)) <= 1e-12 * max(np.max(np.abs(r)), 1e-300), (rank, name)
        for f, v in out.items():
            assert np.max(np.abs(v - hw.get(f)[gidx])) <= 1e-12 * np.max(np.abs(hw.get(f))), (rank, f)
    assert np.count_nonzero(ref[2]) > 100


def test_library_exchange_path_one_rank():
    """The exchange inside the library (sz_comm_init / sz_tile_setup / sz_tile_run: box gather, per-pair slots, pack,
    unpack, drift reference) with one rank -- all of it but the RCCL calls themselves, which need one GPU per rank --
    against the single-context run over more steps than one gather interval: bit-equal columns."""
    import subzero_jl_amd
    from subzero_jl_amd import fields, tiles
    cfg = fields.make_config(n_floes=800, seed=51)
    tw = tiles.TiledWorld(cfg, 0, 1, 0, None, backend="library", rebox_every=20)
    tw.run(45, 0, cfg["dt"], coupling_dt=1)
    hw = fields.build_world(subzero_jl_amd.World(0), cfg)
    hw.run(45, 0, cfg["dt"], coupling_dt=1, stop_on_tags=False)
    for f in FIELDS:
        assert np.array_equal(tw.owned(f), hw.get(f)), f


def test_halo_drift_is_an_error_not_a_missed_contact():
    """floes that out-run the drift margin between two box gathers: the pack kernel raises the halo-drift bit one step
    before a neighbour across a tile edge could be missed"""
    from subzero_jl_amd import fields, tiles
    from subzero_jl_amd.capi import SzError
    cfg = fields.make_config(n_floes=400, seed=52)
    cfg["u"] = np.abs(cfg["u"]) * 40.0 + 4.0                     # 4 .. 8 m/s: 80 .. 160 m per step
    tw = tiles.TiledWorld(cfg, 0, 1, 0, None, backend="library", drift_margin=1500.0, rebox_every=-50)     # (< 0: a FIXED interval)
    tw.run(4, 0, cfg["dt"], coupling_dt=1)                       # 4 steps: at most 640 m, 2 x 640 < 1500
    with pytest.raises(SzError, match="halo-drift"):
        tw.run(30, 4, cfg["dt"], coupling_dt=1)
    # a gather interval that fits the speed is fine
    tw2 = tiles.TiledWorld(cfg, 0, 1, 0, None, backend="library", drift_margin=1500.0, rebox_every=-3)
    tw2.run(30, 0, cfg["dt"], coupling_dt=1)
    # and left to itself the library finds one: the interval follows the measured displacement and the velocities
    tw3 = tiles.TiledWorld(cfg, 0, 1, 0, None, backend="library", drift_margin=1500.0, rebox_every=50)
    tw3.run(60, 0, cfg["dt"], coupling_dt=1)


def test_rccl_binding_self_test():
    """what of the library's RCCL path can run on a one-GPU box: run-time binding, a communicator of one rank, all-gather,
    all-reduce, grouped send / receive to self on the communication stream -- with and without an RCCL already in the process"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n%s"
            "import subzero_jl_amd; w = subzero_jl_amd.World(0); rc = w.L.sz_comm_selftest(w.h); "
            "print('rc', rc, w.L.sz_last_error(w.h).decode()); sys.exit(0 if rc == 0 else 1)")
    for pre in ("", "import torch, torch.distributed\n"):
        out = subprocess.run([sys.executable, "-c", code % (root, pre)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr


# ---------------------------------------------------------------- batches of a tiled run end where the reference runs simplify_floes!
def _tag_cfg(kind, coupled=False):
    """the scenarios of test_hip_parity.py::test_resident_batch_stops_when_a_floe_is_tagged, laid across the edge between two tiles
    (x = 5e4): `fuse` -- two floes owned by DIFFERENT ranks close in until their overlap passes floe_floe_max_overlap; `open` -- a
    floe of rank 1 drifts into an open boundary while rank 0 knows nothing of it"""
    from subzero_jl_amd import floe as floe_mod
    sq = lambda x0, y0, s=1e4: np.array([[x0, y0], [x0, y0 + s], [x0 + s, y0 + s], [x0 + s, y0], [x0, y0]])
    if kind == "fuse":
        kinds = ["periodic"] * 4
        rings = [sq(4.2e4, 4.5e4), sq(4.61e4, 4.6e4), sq(8.0e4, 1.0e4), sq(9.6e4, 6.0e4), sq(1.0e4, 1.0e4)]      # 0 | 1, 2, 3 (has a ghost) | 0
        u = [3.0, -3.0, 0.0, 0.0, 0.0]
    else:
        kinds = ["open"] * 4
        rings = [sq(2.0e4, 4.5e4), sq(8.9e4 + 380.0, 2.0e4), sq(5.2e4, 7.0e4), sq(3.0e4, 1.0e4)]
        u = [0.0, 20.0, 0.0, 0.0]
    n = len(rings)
    off = np.zeros(n + 1, np.int32); off[1:] = np.cumsum([len(r) for r in rings])
    vx = np.concatenate([r[:, 0] for r in rings]); vy = np.concatenate([r[:, 1] for r in rings])
    h = np.full(n, 0.5)
    z = np.zeros((11, 11))
    cfg = dict(n_floes=n, L=1e5, kinds=kinds, vert_off=off, vx=vx, vy=vy, height=h, u=np.array(u), v=np.zeros(n), xi=np.zeros(n), dt=10,
               Nx=10, Ny=10, uo=z, vo=z, hf=z, ua=z, va=z, topography=[], E=1e3, derived=floe_mod.derive(off, vx, vy, h),
               sub_off=np.zeros(n + 1, np.int32), sx=np.zeros(0), sy=np.zeros(0), seed=0)
    if coupled:
        # one-way coupling every step: sub-floe points and a weak ocean that varies in space, so that fxOA .. hflx change from step to step
        # (too weak to move the tag to another step: 1e-3 m/s against closing speeds of 3 - 20 m/s)
        from subzero_jl_amd import fields
        xs = np.linspace(0.0, 1e5, 11)
        cfg["uo"] = 1e-3 * np.sin(2 * np.pi * xs / 1e5)[:, None] * np.ones((1, 11)); cfg["vo"] = 1e-3 * np.ones((11, 1)) * np.cos(2 * np.pi * xs / 1e5)[None, :]
        d = cfg["derived"]; so = np.zeros(n + 1, np.int32); sxs = []; sys_ = []
        for i in range(n):
            sx, sy = fields.subgrid_points(rings[i], d["cx"][i], d["cy"][i], 2.5e3)
            so[i + 1] = so[i] + len(sx); sxs.append(sx); sys_.append(sy)
        cfg["sub_off"] = so; cfg["sx"] = np.concatenate(sxs); cfg["sy"] = np.concatenate(sys_)
    return cfg


def _worker_tags(rank, world, port, kind, q, coupled=False):
    import torch.distributed as dist
    from subzero_jl_amd import tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        cfg = _tag_cfg(kind, coupled)
        tw = tiles.TiledWorld(cfg, rank, world, 0, dist, host_staging=True, backend="library-host", rebox_every=3, drift_margin=3000.0)
        done = tw.run(12, 0, 10, coupling_dt=1 if coupled else 10, coupling_on=coupled, stop_on_tags=True)
        out = {f: tw.owned(f) for f in FIELDS}
        st = tw.world.ids()[2][:len(tw.gidx)]
        fuse = [list(map(int, f)) for f in tw.world.fuse()][:len(tw.gidx)]
        again = tw.run(5, done, 10, coupling_dt=1 if coupled else 10, coupling_on=coupled, stop_on_tags=True)       # tagged at entry: one step
        q.put((rank, tw.gidx, out, done, st, fuse, again))
    finally:
        dist.destroy_process_group()


def _run_worker_tags(*a):
    _guard(_worker_tags)(*a)


@pytest.mark.parametrize("kind,coupled", [("fuse", False), ("open", False), ("fuse", True), ("open", True)])
def test_tiled_batch_ends_on_every_rank_where_a_floe_is_tagged(kind, coupled):
    """The reference runs simplify_floes! after EVERY step (simulation.jl:205-214).  A tiled batch ends after the step in which ANY
    rank tags a floe -- the stop request rides in the header records of the next exchange to every rank, whose unpack kernel ends the
    batch before that step has touched anything -- so both ranks report the same number of steps, and their state, status tags and
    status.fuse_idx (in global floe numbers) are the single context's after that step.  `coupled`: one-way coupling every step -- with peers
    the forcings run beside the exchange, i.e. before a rank knows that the batch has ended, into a second set of output columns; the
    ranks must come back with fxOA .. hflx of the step the batch ended with, not of the one that never ran."""
    import torch.multiprocessing as mp
    import subzero_jl_amd
    from subzero_jl_amd import fields
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_run_worker_tags, args=(r, 2, port, kind, q, coupled)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, 2)
        for p in procs:
            p.join(60)
        assert all(p.exitcode == 0 for p in procs)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    cfg = _tag_cfg(kind, coupled)
    hw = fields.build_world(subzero_jl_amd.World(0), cfg)
    k = hw.run(12, 0, 10, coupling_dt=1 if coupled else 10, coupling_on=coupled)
    assert 2 <= k < 12                                     # the tag comes in the middle of the batch
    tags = hw.ids()[2]; ref_fuse = [list(map(int, f)) for f in hw.fuse()]
    assert np.any(tags[:cfg["n_floes"]] != 1)
    owners = set()
    for rank, gidx, out, done, st, fuse, again in res:
        assert done == k and again == 1, (rank, done, k, again)
        assert np.array_equal(st, tags[gidx]), (rank, st, tags[gidx])
        for f in FIELDS:
            assert np.array_equal(out[f], hw.get(f)[gidx]), (rank, f)
        for i, g in enumerate(gidx):
            assert sorted(fuse[i]) == sorted(ref_fuse[g]), (rank, g, fuse[i], ref_fuse[g])       # (parents only in these scenarios: global = the single context's numbers)
        if np.any(st != 1):
            owners.add(rank)
    if kind == "fuse":
        assert owners == {0, 1}            # the fused pair straddles the tile edge


# ---------------------------------------------------------------- the pause for the largest narrow variant, agreed between the ranks
def _retry_cfg(x0):
    """test_hip_parity.py::_retry_scenario as a tile configuration: two 8-spike stars drifting into each other (6, then 8, then 16 crossings: from
    the sixth step on the pair outgrows the small narrow-phase working set), centres at x0 and x0 + 5 km; two squares on the other tile"""
    from subzero_jl_amd import floe as floe_mod
    th = np.arange(16) * (2 * np.pi / 16)
    rad = np.where(np.arange(16) % 2 == 0, 1.0e4, 0.55e4)
    def star(rot, cx):
        r = np.stack([cx + rad * np.cos(-th + rot), 5e4 + rad * np.sin(-th + rot)], 1)
        return np.concatenate([r, r[:1]])
    sq = lambda x0_, y0, s=1e4: np.array([[x0_, y0], [x0_, y0 + s], [x0_ + s, y0 + s], [x0_ + s, y0], [x0_, y0]])
    rings = [star(0.0, x0), star(np.pi / 8, x0 + 0.5e4), sq(8.0e4, 1.0e4), sq(9.6e4, 6.0e4)]      # the last one has a ghost
    n = len(rings)
    off = np.zeros(n + 1, np.int32); off[1:] = np.cumsum([len(r) for r in rings])
    vx = np.concatenate([r[:, 0] for r in rings]); vy = np.concatenate([r[:, 1] for r in rings])
    h = np.full(n, 0.5); z = np.zeros((11, 11))
    return dict(n_floes=n, L=1e5, kinds=["periodic"] * 4, vert_off=off, vx=vx, vy=vy, height=h, u=np.array([0.0, -30.0, 0.0, 0.0]), v=np.zeros(n), xi=np.zeros(n),
                dt=10, Nx=10, Ny=10, uo=z, vo=z, hf=z, ua=z, va=z, topography=[], E=1e3, derived=floe_mod.derive(off, vx, vy, h),
                sub_off=np.zeros(n + 1, np.int32), sx=np.zeros(0), sy=np.zeros(0), seed=0)


def _worker_retry(rank, world, port, x0, q):
    import torch.distributed as dist
    from subzero_jl_amd import tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        cfg = _retry_cfg(x0)
        tw = tiles.TiledWorld(cfg, rank, world, 0, dist, host_staging=True, backend="library-host", rebox_every=3, drift_margin=3000.0)
        done = tw.run(8, 0, 10, coupling_dt=10, coupling_on=False)
        nretry = int(tw.world.stats()["n_retry"])
        done2 = tw.run(2, 8, 10, coupling_dt=10, coupling_on=False)
        assert tw.world.crec_mismatches() == 0          # (the collision records of the owned floes are the columns', also after the restart)
        q.put((rank, tw.gidx, {f: tw.owned(f) for f in FIELDS}, done, done2, nretry))
    finally:
        dist.destroy_process_group()


def _run_worker_retry(*a):
    _guard(_worker_retry)(*a)


@pytest.mark.parametrize("x0", [4.0e4, 4.7e4])
def test_tiled_batch_pauses_for_the_largest_narrow_variant(x0):
    """sz_tile_run leaves the largest narrow-phase variant out of its steps, as sz_step does.  When a pair outgrows the small working set in the
    middle of a batch, the rank that holds it pauses inside that step; the pause reaches the other rank with the next exchange, which stops
    before that step has touched anything; the first rank finishes its step and both run the rest of the batch again.  x0 = 40 km: both stars
    on rank 0, far from rank 1's halo (only rank 0 pauses); x0 = 47 km: the pair straddles the tile edge and both ranks meet it.  Owned columns
    bit-equal to the single context (which pauses the same way)."""
    import torch.multiprocessing as mp
    import subzero_jl_amd
    from subzero_jl_amd import fields
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_run_worker_retry, args=(r, 2, port, x0, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, 2)
        for p in procs:
            p.join(60)
        assert all(p.exitcode == 0 for p in procs)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    cfg = _retry_cfg(x0)
    hw = fields.build_world(subzero_jl_amd.World(0), cfg)
    assert hw.run(8, 0, 10, coupling_dt=10, coupling_on=False, stop_on_tags=False) == 8
    assert hw.stats()["n_retry"] >= 1
    assert hw.run(2, 8, 10, coupling_dt=10, coupling_on=False, stop_on_tags=False) == 2
    paused = set()
    for rank, gidx, out, done, done2, nretry in res:
        assert done == 8 and done2 == 2, (rank, done, done2)
        if nretry:
            paused.add(rank)
        for f in FIELDS:
            assert np.array_equal(out[f], hw.get(f)[gidx]), (rank, f)
    assert paused == ({0} if x0 < 4.5e4 else {0, 1}), paused


def _pause_and_tag_cfg(w0):
    """the two stars of _retry_cfg on the left tile (their pair needs the largest narrow variant a few steps in) and, on the right tile, two
    squares closing in on each other from an initial overlap of w0 metres until they fuse (collisions.jl:366)"""
    from subzero_jl_amd import floe as floe_mod
    cfg = _retry_cfg(4.0e4)
    sq = lambda x0_, y0, s=1e4: np.array([[x0_, y0], [x0_, y0 + s], [x0_ + s, y0 + s], [x0_ + s, y0], [x0_, y0]])
    off = cfg["vert_off"]
    rings = [np.stack([cfg["vx"][off[k]:off[k + 1]], cfg["vy"][off[k]:off[k + 1]]], 1) for k in range(2)]
    rings += [sq(6.0e4, 1.5e4), sq(7.0e4 - w0, 1.6e4)]
    n = len(rings)
    off = np.zeros(n + 1, np.int32); off[1:] = np.cumsum([len(r) for r in rings])
    vx = np.concatenate([r[:, 0] for r in rings]); vy = np.concatenate([r[:, 1] for r in rings])
    cfg.update(n_floes=n, vert_off=off, vx=vx, vy=vy, height=np.full(n, 0.5), u=np.array([0.0, -30.0, 3.0, -3.0]), v=np.zeros(n), xi=np.zeros(n),
               derived=floe_mod.derive(off, vx, vy, np.full(n, 0.5)), sub_off=np.zeros(n + 1, np.int32))
    return cfg


def _worker_pause_and_tag(rank, world, port, w0, q):
    import torch.distributed as dist
    from subzero_jl_amd import tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        cfg = _pause_and_tag_cfg(w0)
        tw = tiles.TiledWorld(cfg, rank, world, 0, dist, host_staging=True, backend="library-host", rebox_every=3, drift_margin=3000.0)
        done = tw.run(10, 0, 10, coupling_dt=10, coupling_on=False, stop_on_tags=True)
        q.put((rank, tw.gidx, {f: tw.owned(f) for f in FIELDS + ["status"]}, done, int(tw.world.stats()["n_retry"])))
    finally:
        dist.destroy_process_group()


def _run_worker_pause_and_tag(*a):
    _guard(_worker_pause_and_tag)(*a)


def test_a_pause_on_one_rank_and_a_tag_on_another_in_the_same_step():
    """Rank 0's narrow phase meets an item for the largest variant in step k (it pauses inside that step) while rank 1 tags a floe `fuse` in the
    very same step (the batch ends after it).  Neither rank hears of the other's word through the halo headers -- the unpack kernels of step
    k + 1 return at their own stop test first -- so the ranks agree on both words before anyone branches (comm_agree_steps): rank 0 finishes
    its step, both return after step k with the single context's state.  The overlap of the fusing pair is tuned (with the single context) so
    that the two events fall into one step."""
    import torch.multiprocessing as mp
    import subzero_jl_amd
    from subzero_jl_amd import fields

    def events(w0):
        hw = fields.build_world(subzero_jl_amd.World(0), _pause_and_tag_cfg(w0))
        pause = fuse = None
        for k in range(12):
            d = hw.run(1, k, 10, coupling_dt=10, coupling_on=False, stop_on_tags=True)
            if pause is None and hw.stats()["n_retry"] >= 1:
                pause = k + 1
            if np.any(hw.get("status")[2:] != 1):
                fuse = k + 1
                break
        return pause, fuse
    w0 = None
    for cand in range(5900, 6260, 15):
        p_, f_ = events(float(cand))
        if p_ is not None and f_ == p_:
            w0 = float(cand); k = p_
            break
    assert w0 is not None, "no overlap puts the fuse into the pause's step"
    assert 1 <= k <= 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_run_worker_pause_and_tag, args=(r, 2, port, w0, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, 2)
        for p in procs:
            p.join(60)
        assert all(p.exitcode == 0 for p in procs)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    hw = fields.build_world(subzero_jl_amd.World(0), _pause_and_tag_cfg(w0))
    assert hw.run(10, 0, 10, coupling_dt=10, coupling_on=False, stop_on_tags=True) == k
    paused = set(); tagged = set()
    for rank, gidx, out, done, nretry in res:
        assert done == k, (rank, done, k)
        if nretry:
            paused.add(rank)
        if np.any(out["status"] != 1):
            tagged.add(rank)
        for f in FIELDS + ["status"]:
            assert np.array_equal(out[f], hw.get(f)[gidx]), (rank, f)
    assert paused == {0} and tagged == {1}, (paused, tagged)


def _pause_in_a_fast_field_cfg(seed):
    """fast floes that cross the periodic walls all the time (parents swap with their ghosts: collisions.jl:942-950) around the two stars whose
    pair needs the largest narrow variant a few steps in (the pause), which sit in the middle of the left tile"""
    from subzero_jl_amd import floe as floe_mod
    base = _field(700, seed, fast=True)
    L = base["L"]; off = base["vert_off"]
    th = np.arange(16) * (2 * np.pi / 16)
    rad = np.where(np.arange(16) % 2 == 0, 1.0e4, 0.55e4) * (L / 4.0e5)          # (the stars of _retry_cfg, scaled to this box)
    c0 = np.array([0.25 * L, 0.5 * L])
    def star(rot, cx):
        r = np.stack([cx + rad * np.cos(-th + rot), c0[1] + rad * np.sin(-th + rot)], 1)
        return np.concatenate([r, r[:1]])
    stars = [star(0.0, c0[0]), star(np.pi / 8, c0[0] + 0.5 * rad.max())]
    keep = [k for k in range(base["n_floes"]) if np.hypot(base["derived"]["cx"][k] - c0[0] - 0.25 * rad.max(), base["derived"]["cy"][k] - c0[1]) > 3.2 * rad.max()]
    rings = stars + [np.stack([base["vx"][off[k]:off[k + 1]], base["vy"][off[k]:off[k + 1]]], 1) for k in keep]
    n = len(rings)
    o2 = np.zeros(n + 1, np.int32); o2[1:] = np.cumsum([len(r) for r in rings])
    vx = np.concatenate([r[:, 0] for r in rings]); vy = np.concatenate([r[:, 1] for r in rings])
    h = np.concatenate([[0.5, 0.5], base["height"][keep]])
    cfg = dict(base)
    cfg.update(n_floes=n, vert_off=o2, vx=vx, vy=vy, height=h, u=np.concatenate([[0.0, -0.75 * rad.max() / (6 * base["dt"])], base["u"][keep]]),
               v=np.concatenate([[0.0, 0.0], base["v"][keep]]), xi=np.concatenate([[0.0, 0.0], base["xi"][keep]]),
               derived=floe_mod.derive(o2, vx, vy, h), sub_off=np.zeros(n + 1, np.int32), sx=np.zeros(0), sy=np.zeros(0))
    return cfg


def _worker_pause_fast(rank, world, port, seed, steps, q):
    import torch.distributed as dist
    from subzero_jl_amd import tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        cfg = _pause_in_a_fast_field_cfg(seed)
        tw = tiles.TiledWorld(cfg, rank, world, 0, dist, host_staging=True, backend="library-host", rebox_every=3)
        done = tw.run(steps, 0, cfg["dt"], coupling_dt=10, coupling_on=False)
        off, x, y = tw.world.rings()
        q.put((rank, tw.gidx, {f: tw.owned(f) for f in FIELDS}, done, int(tw.world.stats()["n_retry"]), x[:off[len(tw.gidx)]].copy()))
    finally:
        dist.destroy_process_group()


def _run_worker_pause_fast(*a):
    _guard(_worker_pause_fast)(*a)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_a_pause_among_parents_that_swap_with_their_ghosts(seed):
    """The pause for the largest narrow variant (rank 0's star pair) in a field whose parents cross the periodic walls all the time.  The ranks
    that did NOT pause have the paused step behind them like any other -- their integrator has made the next ghosts from the parents
    BEFORE their swap (collisions.jl:942-950) -- and take the steps up where they stopped; they used to start again from parents already
    swapped, which re-numbers the ghosts of a corner parent and rounds a coordinate once more.  Two ranks, owned columns and ring points
    bit-equal to the single context."""
    import torch.multiprocessing as mp
    import subzero_jl_amd
    from subzero_jl_amd import fields
    steps = 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_run_worker_pause_fast, args=(r, 2, port, seed, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, 2)
        for p in procs:
            p.join(60)
        assert all(p.exitcode == 0 for p in procs)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    cfg = _pause_in_a_fast_field_cfg(seed)
    hw = fields.build_world(subzero_jl_amd.World(0), cfg)
    assert hw.run(steps, 0, cfg["dt"], coupling_dt=10, coupling_on=False, stop_on_tags=False) == steps
    assert hw.stats()["n_retry"] >= 1
    wrapped = np.abs(hw.get("cx") - cfg["derived"]["cx"]) > 0.5 * cfg["L"]
    assert wrapped.sum() >= 3
    hoff, hx, _ = hw.rings()
    paused = set()
    for rank, gidx, out, done, nretry, vx in res:
        assert done == steps
        if nretry:
            paused.add(rank)
        for f in FIELDS:
            assert np.array_equal(out[f], hw.get(f)[gidx]), (rank, f, np.max(np.abs(out[f] - hw.get(f)[gidx])))
        assert np.array_equal(vx, np.concatenate([hx[hoff[g]:hoff[g + 1]] for g in gidx])), rank
    assert 0 in paused


def _many_vertex_cfg(seed=5, n_side=12):
    """a periodic field of star polygons with 34 .. 60 vertices each (rings far above the 32 points a halo record used to hold; the reference's
    own fixture outlines have 35 .. 203 points, Floe rings are unbounded: floe.jl:24-77), overlapping their neighbours, with some drift"""
    from subzero_jl_amd import floe as floe_mod
    rng = np.random.default_rng(seed)
    L = 1.2e5; sp = L / n_side
    rings = []
    for gy in range(n_side):
        for gx in range(n_side):
            nv = int(rng.integers(34, 61))
            th = ((2 * np.pi / nv) * (np.arange(nv) + rng.uniform(-0.3, 0.3, nv)))[::-1]
            rad = 0.62 * sp * rng.uniform(0.7, 1.0, nv)
            cx = (gx + 0.5) * sp + rng.uniform(-0.1, 0.1) * sp; cy = (gy + 0.5) * sp + rng.uniform(-0.1, 0.1) * sp
            r = np.stack([cx + rad * np.cos(th), cy + rad * np.sin(th)], 1)
            rings.append(np.vstack([r, r[:1]]))
    n = len(rings)
    off = np.zeros(n + 1, np.int32); off[1:] = np.cumsum([len(r) for r in rings])
    vx = np.concatenate([r[:, 0] for r in rings]); vy = np.concatenate([r[:, 1] for r in rings])
    h = np.full(n, 0.3); z = np.zeros((11, 11))
    return dict(n_floes=n, L=L, kinds=["periodic"] * 4, vert_off=off, vx=vx, vy=vy, height=h, u=rng.uniform(-0.3, 0.3, n), v=rng.uniform(-0.3, 0.3, n),
                xi=rng.uniform(-1e-5, 1e-5, n), dt=10, Nx=10, Ny=10, uo=z, vo=z, hf=z, ua=z, va=z, topography=[], E=6e6, derived=floe_mod.derive(off, vx, vy, h),
                sub_off=np.zeros(n + 1, np.int32), sx=np.zeros(0), sy=np.zeros(0), seed=seed)


def _worker_many_vertex(rank, world, port, steps, q):
    import torch.distributed as dist
    from subzero_jl_amd import tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        cfg = _many_vertex_cfg()
        tw = tiles.TiledWorld(cfg, rank, world, 0, dist, host_staging=True, backend="library-host", rebox_every=3)
        assert tw.REC >= 12 + 2 * 60                      # the records follow the largest ring of all ranks
        done = tw.run(steps, 0, cfg["dt"], coupling_dt=10, coupling_on=False)
        q.put((rank, tw.gidx, {f: tw.owned(f) for f in FIELDS}, done, tw.n_halo_last))
    finally:
        dist.destroy_process_group()


def _run_worker_many_vertex(*a):
    _guard(_worker_many_vertex)(*a)


@pytest.mark.parametrize("world", [2, 4])
def test_tiles_take_rings_above_32_points(world):
    """A tiled run used to refuse rings above the 32 points its halo records held (ERR_CAP_RING) while the single context takes 255.  The
    records now have room for the largest ring of any rank's floes (sz_tile_enable): a field of 34 .. 60-vertex floes in 2 and 2 x 2 tiles,
    six steps, owned columns bit-equal to the single context."""
    import torch.multiprocessing as mp
    import subzero_jl_amd
    from subzero_jl_amd import fields
    steps = 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_run_worker_many_vertex, args=(r, world, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, world)
        for p in procs:
            p.join(60)
        assert all(p.exitcode == 0 for p in procs)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    cfg = _many_vertex_cfg()
    assert np.diff(cfg["vert_off"]).min() > 32
    hw = fields.build_world(subzero_jl_amd.World(0), cfg)
    assert hw.run(steps, 0, cfg["dt"], coupling_dt=10, coupling_on=False, stop_on_tags=False) == steps
    assert hw.stats()["n_pairs"] > cfg["n_floes"]
    seen = np.zeros(cfg["n_floes"], bool)
    for rank, gidx, out, done, nhalo in res:
        assert done == steps and nhalo > 0
        seen[gidx] = True
        for f in FIELDS:
            assert np.array_equal(out[f], hw.get(f)[gidx]), (rank, f, np.max(np.abs(out[f] - hw.get(f)[gidx])))
    assert seen.all()


# ---------------------------------------------------------------- migration inside the library
def _worker_migrate(rank, world, port, n, seed, steps, every, q, shape="star", fast=True, stop=False):
    import time
    import torch.distributed as dist
    from subzero_jl_amd import tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        cfg = _field(n, seed, fast=fast, shape=shape)
        backend = os.environ.get("SZ_FUZZ_BACKEND", "library-host")          # ("torch": the older path of one collective per step on buffers the library packs)
        tw = tiles.TiledWorld(cfg, rank, world, 0, dist, host_staging=True, backend=backend, rebox_every=5 if backend != "torch" else 50)
        if os.environ.get("SZ_FUZZ_PRECISION") == "mixed":          # (tools/fuzz_tiles.py: fp32 forcings and broad-phase records, fp64 world rings)
            tw.world.set_precision("mixed")
        moved, cost, ran = 0, [], 0
        for t0 in range(0, steps, every):
            k = min(every, steps - t0)
            done = tw.run(k, t0, cfg["dt"], coupling_dt=1, stop_on_tags=stop)
            ran += done
            if done < k:               # a floe was tagged: the batch ended on every rank (the host's simplify_floes! would run now)
                break
            if stop:                   # ... or it was tagged in the batch's last step: the host looks at the tags after every batch
                tags = [None] * world
                dist.all_gather_object(tags, bool(np.any(tw.owned("status") != 1)))
                if any(tags):
                    break
            if t0 + every < steps and not os.environ.get("SZ_PROBE_SKIP_MIGRATE"):          # (the switch: tools/probe/tiles_first_diff.py)
                if backend == "torch":
                    moved += tw.repartition()
                    continue
                t = time.perf_counter(); moved += tw.migrate(); cost.append(time.perf_counter() - t)
                assert tw.migrate_path == 1 or os.environ.get("SZ_PROBE_ANY_PATH"), tw.migrate_path          # floes that drifted over a tile edge: packed and placed on the device
        out = {f: tw.owned(f) for f in FIELDS}
        off, x, y = tw.world.rings()
        k = off[len(tw.gidx)]
        out["_rings"] = (off[:len(tw.gidx) + 1].copy(), x[:k].copy(), y[:k].copy())
        out["_ran"] = ran
        q.put((rank, tw.gidx, out, moved, cost))
    finally:
        dist.destroy_process_group()


def _run_worker_migrate(*a):
    _guard(_worker_migrate)(*a)


@pytest.mark.parametrize("world,n", [(2, 500), (4, 1000)])
def test_migration_every_20_steps(world, n):
    """sz_tile_migrate: fast floes (2 - 8 m/s: 40 - 160 m per step on tiles a few hundred km wide, parents
    crossing the periodic walls) with a re-tile every 20 steps -- floes that left their tile go to their new owner with their complete
    state over the library's channel, packed and placed on the device (csrc/sz_migrate.hpp) -- and after 60 steps every owned column is
    bit-equal to the single context's.  The cost per re-tile is printed."""
    migration_case(world, n, 78, 60, 20)


@pytest.mark.parametrize("world,n,seed,steps,every", [(2, 573, 5112, 25, 10), (4, 2336, 5090, 27, 9), (2, 703, 5056, 54, 13)])
def test_ghosts_of_a_halo_floe_in_the_step_it_swaps(world, n, seed, steps, every):
    """Cases tools/fuzz_tiles.py found (3 of 150): a corner parent leaves the domain and swaps with its ghost (collisions.jl:942-950) while it is a
    halo floe of another rank.  In the step of the swap the reference numbers the ghosts as made from the parent BEFORE the swap; ghosts made from the
    parent after it are the same four places with two numbers exchanged and one coordinate rounded once more, the Dict rule (collisions.jl:751-775)
    then keeps another instance pair of the same contact, and the rows differ in the last bits.  The halo record therefore carries the floe as the
    update left it, before the swap, and the receiving rank swaps it itself: bit-equal to the single context, rings included."""
    migration_case(world, n, seed, steps, every, verbose=False)


@pytest.mark.parametrize("world,n,seed,steps,every,shape,fast", [(4, 2006, 30025, 40, 9, "walls-topo", True), (4, 1943, 30071, 25, 12, "walls", False),
                                                                 (2, 977, 30003, 51, 11, "walls-topo", False)])
def test_tiles_between_collision_walls(world, n, seed, steps, every, shape, fast):
    """Fields between collision walls (configs[3]: no ghosts, floe - wall and floe - topography items) in tiles, cases of tools/fuzz_tiles.py ... walls.
    The first one pauses for the largest narrow variant while the forcings of a spatially varying ocean run beside the exchange: the integrator that
    finishes the paused step must read the forcing outputs of THAT step (the steps enqueued behind it go on alternating the two output sets on the
    host).  The others ran into the element scan riding in the tail of a tile's neighbour launch, which commits the halo rows' count while the scan
    reads it (spurious row-capacity errors and last-digit differences, from run to run): a tile's element items have a launch of their own."""
    migration_case(world, n, seed, steps, every, verbose=False, shape=shape, fast=fast)


def migration_case(world, n, seed, steps, every, verbose=True, shape="star", fast=True, stop=False):
    """`world` ranks sharing the GPU run `steps` steps of the field (n, seed; fast: floes at 40 x the usual speed, shifted across the walls) with a
    re-tile every `every` steps; every owned column and ring point must equal the single context's.  stop: batches end after the first step that tags
    a floe, on every rank and in the single context at the same step (also used by tools/fuzz_tiles.py)"""
    import torch.multiprocessing as mp
    import subzero_jl_amd
    from subzero_jl_amd import fields
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_run_worker_migrate, args=(r, world, port, n, seed, steps, every, q, shape, fast, stop)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, world)
        for p in procs:
            p.join(60)
        assert all(p.exitcode == 0 for p in procs)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    cfg = _field(n, seed, fast=fast, shape=shape)
    hw = fields.build_world(subzero_jl_amd.World(0), cfg)
    if os.environ.get("SZ_FUZZ_PRECISION") == "mixed":
        hw.set_precision("mixed")          # (with SZ_BODY_RINGS=0 in the environment: a tile keeps its rings in world coordinates, fp64)
    ran = hw.run(steps, 0, cfg["dt"], coupling_dt=1, stop_on_tags=stop)
    seen = np.zeros(n, bool); moved = 0
    for rank, gidx, out, mv, cost in res:
        assert out["_ran"] == ran, (rank, "steps run", out["_ran"], ran)
        assert not seen[gidx].any()
        seen[gidx] = True; moved += mv
        if verbose:
            print(f"rank {rank}: {len(gidx)} floes owned at the end, {mv} given away, re-tile cost {['%.1f ms' % (1e3 * c) for c in cost]}")
        for f in FIELDS:
            assert np.array_equal(out[f], hw.get(f)[gidx]), (rank, f, np.max(np.abs(out[f] - hw.get(f)[gidx])))
        off, x, y = out["_rings"]; hoff, hx, hy = hw.rings()           # the rings too, point by point
        for k, g in enumerate(gidx):
            assert np.array_equal(x[off[k]:off[k + 1]], hx[hoff[g]:hoff[g + 1]]) and np.array_equal(y[off[k]:off[k + 1]], hy[hoff[g]:hoff[g + 1]]), (rank, "ring", int(g))
    assert seen.all() and (moved > 0 or os.environ.get("SZ_PROBE_SKIP_MIGRATE") or not fast or ran < steps or shape != "star")
    return moved


def _worker_overflow(rank, world, port, q):
    import torch.distributed as dist
    from subzero_jl_amd import tiles
    from subzero_jl_amd.capi import SzError
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        cfg = _field(4000, 35)
        # a drift margin as wide as the domain: every floe of every other rank is in this rank's halo -- 3 000 floes and their ghosts for a tile whose
        # upload of 1 000 floes carved max(2 M + 64, M + 2048) = 3 048 rows
        tw = tiles.TiledWorld(cfg, rank, world, 0, dist, host_staging=True, backend="library-host", rebox_every=5, drift_margin=float(cfg["L"]))
        try:
            tw.run(4, 0, cfg["dt"], coupling_dt=1)
            q.put((rank, "no error"))
        except SzError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def _run_worker_overflow(*a):
    _guard(_worker_overflow)(*a)


def test_a_halo_that_outgrows_the_tile_is_an_error_on_every_rank():
    """More halo floes than the tile has rows for: the unpack kernel refuses the rows it has no room for, marks the step's allocator
    poisoned (the neighbour search then runs on the owned floes alone instead of walking rows nobody wrote) and raises the capacity
    bit -- which all ranks agree on at the end of the batch: every rank raises, none hangs, nothing is written out of bounds."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_run_worker_overflow, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, 4)
        for p in procs:
            p.join(60)
        assert all(p.exitcode == 0 for p in procs)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    for rank, msg in res:
        assert "error bits" in msg and msg != "no error", (rank, msg)
