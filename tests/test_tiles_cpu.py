"""world_size-2 (and 4) gloo tests of the multi-GPU protocol on the CPU: tile assignment, the halo
selection rule and the all-to-all-v exchange, with the CPU oracle standing in for the kernels.
Claim under test: running the ordinary single-process collision path on (owned + halo) floes,
ordered by global index, gives every OWNED floe bit-identical contact rows and totals to the
global run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import subzero_jl_amd
from subzero_jl_amd import fields, tiles
from oracle import orc


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _oracle_world(cfg, idx):
    """oracle world holding floes `idx` (sorted global indices), ids = global index + 1"""
    w = orc.World()
    w.set_consts(E=cfg["E"]); w.set_settings()
    w.set_domain([fields.KIND[k] for k in cfg["kinds"]], 0.0, cfg["L"], 0.0, cfg["L"])
    off, vx, vy = cfg["vert_off"], cfg["vx"], cfg["vy"]
    for i in idx:
        w.add_floe(np.stack([vx[off[i]:off[i + 1]], vy[off[i]:off[i + 1]]], 1), cfg["height"][i])
    w.set("u", cfg["u"][idx]); w.set("v", cfg["v"][idx]); w.set("xi", cfg["xi"][idx])
    w.set_ids(np.asarray(idx, np.int64) + 1)
    return w


def _worker(rank, world, port, n, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = fields.make_config(n_floes=n, seed=seed)
        d = cfg["derived"]; L = cfg["L"]
        owner = tiles.assign_tiles(d["cx"], d["cy"], L, world)
        own = np.nonzero(owner == rank)[0]
        # --- owned boxes, all-gathered
        b5 = torch.tensor([d["cx"][own].min(), d["cx"][own].max(), d["cy"][own].min(), d["cy"][own].max(),
                           d["rmax"][own].max()], dtype=torch.float64)
        allb = [torch.zeros(5, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allb, b5)
        allb = np.stack([b.numpy() for b in allb]); allb[:, 4] = allb[:, 4].max()
        boxes = [tiles.expanded_box(b, 0.0) for b in allb]
        # --- pack rule + all-to-all-v of the records (here: the global index is the record)
        send = [own[tiles.select_halo(d["cx"][own], d["cy"][own], boxes[r], L, L, True, True)] if r != rank
                else np.zeros(0, int) for r in range(world)]
        counts = torch.tensor([len(s) for s in send], dtype=torch.int64); rcounts = torch.zeros_like(counts)
        dist.all_to_all_single(rcounts, counts)
        sbuf = torch.tensor(np.concatenate(send).astype(np.float64)); rbuf = torch.zeros(int(rcounts.sum()), dtype=torch.float64)
        dist.all_to_all_single(rbuf, sbuf, [int(c) for c in rcounts], [int(c) for c in counts])
        halo = rbuf.numpy().astype(int)
        assert len(np.intersect1d(halo, own)) == 0 and len(np.unique(halo)) == len(halo)
        local = np.sort(np.concatenate([own, halo]))
        # --- local run vs global run
        lw = _oracle_world(cfg, local); lw.add_ghosts(); lw.timestep_collisions(len(local), cfg["dt"])
        gw = _oracle_world(cfg, np.arange(n)); gw.add_ghosts(); gw.timestep_collisions(n, cfg["dt"])
        pos = np.searchsorted(local, own)
        for f in ("coll_fx", "coll_fy", "coll_trq", "overarea"):
            assert np.array_equal(lw.get(f)[pos], gw.get(f)[own]), f
        lo, lr = lw.interactions(); go, gr = gw.interactions()
        nrows = 0
        for p, g in zip(pos, own):
            a, b = lr[lo[p]:lo[p + 1]], gr[go[g]:go[g + 1]]
            assert a.shape == b.shape and np.array_equal(a[:, 1:], b[:, 1:])
            nrows += len(a)
        q.put((rank, len(own), len(halo), nrows))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,seed", [(2, 300, 21), (4, 600, 22), (8, 1000, 23)])
def test_halo_protocol_gloo(world, n, seed):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = sorted(q.get() for _ in range(world))
    assert sum(r[1] for r in res) == n
    assert all(r[2] > 0 and r[3] > 0 for r in res)       # every rank received a halo and has contacts
    assert all(r[2] < n - r[1] for r in res) or world == 2


def test_tile_grid_shapes():
    assert [tiles.tile_grid(w) for w in (1, 2, 4, 8)] == [(1, 1), (2, 1), (2, 2), (4, 2)]
