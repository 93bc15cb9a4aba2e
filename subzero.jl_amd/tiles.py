"""Multi-GPU sharding of the floe population by spatial tile with a ghost-floe halo.

One process per GPU.  The domain box is cut into px x py tiles; a rank OWNS the floes whose
centroid lies in its tile at partition time and keeps them for the whole run (floes move metres
per step against tile sizes of hundreds of kilometres; `repartition()` is a rare host-side
operation).  Every step each rank

  1. packs, on the device, the owned floes that any other rank may need: those whose centroid --
     or a periodic image of it -- lies inside the other rank's owned bounding box expanded by the
     interaction range 2*max(rmax) (+ a drift margin);
  2. trades them with ONE all-to-all-v (`all_to_all_single` with split sizes: RCCL over xGMI; the
     messages are KB-scale, the exchange is latency-bound);
  3. appends the received records as extra floes and runs the ordinary single-GPU timestep on
     owned + halo floes.  Order-dependent rules (which floe of a pair is p1, the Dict rule, row
     order) use the floes' GLOBAL indices, so every pair is evaluated with the same roles on
     every rank that sees it and the owned floes get bit-identical contact rows to a single-GPU
     run; cross-tile pairs are evaluated on both sides instead of shipping rows back.
  4. Only owned floes are integrated; the halo is dropped and rebuilt next step.

The periodic ghost floes of the reference (collisions.jl:881-1047) are created locally from the
local parents (owned or halo) by the same kernels as in the single-GPU path.
"""
import ctypes as C

import numpy as np

from . import capi
from .host import World


def tile_grid(world):
    """px x py with px >= py and px*py == world: 1x1, 2x1, 2x2, 4x2, ..."""
    py = int(np.floor(np.sqrt(world)))
    while world % py:
        py -= 1
    return world // py, py


def assign_tiles(cx, cy, L, world):
    px, py = tile_grid(world)
    ix = np.clip((cx / L * px).astype(int), 0, px - 1)
    iy = np.clip((cy / L * py).astype(int), 0, py - 1)
    return iy * px + ix


def select_halo(cx, cy, box, Lx, Ly, per_x, per_y):
    """Host-side statement of the pack rule (used by the CPU tests): mask of floes whose centroid
    or one of its periodic images lies inside `box` = (xmin, xmax, ymin, ymax)."""
    hit = np.zeros(len(cx), bool)
    for kx in ((-1, 0, 1) if per_x else (0,)):
        for ky in ((-1, 0, 1) if per_y else (0,)):
            x = cx + kx * Lx; y = cy + ky * Ly
            hit |= (box[0] <= x) & (x <= box[1]) & (box[2] <= y) & (y <= box[3])
    return hit


def expanded_box(b5, margin):
    """b5 = xmin, xmax, ymin, ymax, rmax of the owned floes -> box expanded by 2*rmax + margin."""
    r = 2.0 * b5[4] + margin
    return np.array([b5[0] - r, b5[1] + r, b5[2] - r, b5[3] + r])


def subset_config(cfg, idx):
    """The columns of a fields.make_config() scenario restricted to the floes `idx` (sorted)."""
    off = cfg["vert_off"]; so = cfg["sub_off"]
    nv = np.diff(off)[idx]; ns = np.diff(so)[idx]
    voff = np.zeros(len(idx) + 1, np.int32); voff[1:] = np.cumsum(nv)
    soff = np.zeros(len(idx) + 1, np.int32); soff[1:] = np.cumsum(ns)
    vsel = np.concatenate([np.arange(off[i], off[i + 1]) for i in idx]) if len(idx) else np.zeros(0, int)
    ssel = np.concatenate([np.arange(so[i], so[i + 1]) for i in idx]) if len(idx) else np.zeros(0, int)
    d = cfg["derived"]
    cols = dict(cx=d["cx"][idx], cy=d["cy"][idx], rmax=d["rmax"][idx], area=d["area"][idx], height=d["height"][idx],
                mass=d["mass"][idx], moment=d["moment"][idx], u=cfg["u"][idx], v=cfg["v"][idx], xi=cfg["xi"][idx],
                vert_off=voff, vx=cfg["vx"][vsel], vy=cfg["vy"][vsel],
                id=(np.asarray(idx) + 1).astype(np.int64))
    return cols, soff, cfg["sx"][ssel], cfg["sy"][ssel]


def host_transport(dist, rank, world):
    """sz_host_transport over torch.distributed collectives on HOST tensors (gloo, or any backend that takes CPU tensors):
    the three collectives the library calls when the tiled run was set up with sz_comm_init_host."""
    import torch

    def view(p, nbytes):
        return torch.frombuffer((C.c_char * nbytes).from_address(p), dtype=torch.uint8) if nbytes else torch.zeros(0, dtype=torch.uint8)

    def allgather(user, send, recv, nbytes):
        try:
            out = view(recv, nbytes * world)
            dist.all_gather(list(out.split(nbytes)), view(send, nbytes).clone())
            return 0
        except Exception:                      # noqa: BLE001  (an exception must not cross the C frames)
            import traceback; traceback.print_exc()
            return 1

    def sendrecv(user, npeers, peer, send, send_bytes, recv, recv_bytes):
        try:
            ops = []
            for k in range(npeers):
                if recv_bytes[k]:
                    ops.append(dist.P2POp(dist.irecv, view(recv[k], recv_bytes[k]), int(peer[k])))
                if send_bytes[k]:
                    ops.append(dist.P2POp(dist.isend, view(send[k], send_bytes[k]), int(peer[k])))
            for r in (dist.batch_isend_irecv(ops) if ops else []):
                r.wait()
            return 0
        except Exception:                      # noqa: BLE001
            import traceback; traceback.print_exc()
            return 1

    def allreduce(user, buf, n):
        try:
            dist.all_reduce(torch.frombuffer((C.c_double * n).from_address(C.addressof(buf.contents)), dtype=torch.float64))
            return 0
        except Exception:                      # noqa: BLE001
            import traceback; traceback.print_exc()
            return 1

    t = capi.SzHostTransport(None, capi._AG(allgather), capi._SR(sendrecv), capi._AR(allreduce))
    t._keep = (allgather, sendrecv, allreduce)
    return t


class TileSetupError(capi.SzError):
    """the exchange could not be set up -- raised on EVERY rank together (the ranks agree on it before any collective that
    one of them would miss), so that a caller may fall back to another backend on all ranks at once"""


class TiledWorld:
    """One rank of a tiled run.  `dist` is torch.distributed (initialised) or None for world == 1."""

    def __init__(self, cfg, rank, world, device, dist, drift_margin=None, rebox_every=50, host_staging=False,
                 always_exchange=False, backend="torch"):
        """backend: "library" -- the exchange runs inside libsubzero_hip.so (sz_comm_init / sz_tile_setup / sz_tile_run: RCCL
        bound by the library, grouped send / receive with the neighbouring tiles only; `dist` is then only used to hand the
        communicator id to the ranks); "torch" -- one torch.distributed all_to_all_single per step on buffers the library
        packs and unpacks (and, with host_staging, through the host: the gloo tests); "library-host" -- sz_tile_run as
        with "library", but over the host's channel (sz_comm_init_host: `dist` collectives on host buffers, e.g. gloo) --
        the library's multi-rank exchange without an RCCL communicator, which ranks sharing one GPU cannot open."""
        import torch
        self.torch = torch
        self.cfg, self.rank, self.nranks, self.dist = cfg, rank, world, dist
        self.backend = backend
        self.host_staging = host_staging          # gloo: exchange through CPU tensors
        self.world_device = device
        self.always_exchange = always_exchange    # run the collectives even with one rank (tests)
        self.L = cfg["L"]
        self.per_x = cfg["kinds"][2] == "periodic"; self.per_y = cfg["kinds"][0] == "periodic"
        owner = assign_tiles(cfg["derived"]["cx"], cfg["derived"]["cy"], self.L, world)
        self.gidx = np.nonzero(owner == rank)[0]
        cols, soff, sx, sy = subset_config(cfg, self.gidx)
        from . import fields
        w = World(device)
        w.set_consts(E=cfg["E"]); w.set_settings()
        w.set_domain([fields.KIND[k] for k in cfg["kinds"]], 0.0, self.L, 0.0, self.L)
        if cfg["topography"]:
            w.set_topography(cfg["topography"])
        w.set_grid_fields(cfg["Nx"], cfg["Ny"], 0.0, self.L, 0.0, self.L, cfg["uo"], cfg["vo"], cfg["hf"], cfg["ua"], cfg["va"])
        w.load_columns(cols); w.set_subpoints_csr(soff, sx, sy)
        w._push()
        g = np.ascontiguousarray(self.gidx, np.int64)
        max_ring = float(np.diff(cfg["vert_off"]).max())        # over ALL floes: halo floes arrive unseen
        max_rmax = float(cfg["derived"]["rmax"].max())          # over ALL floes, like max_ring
        self._max_ring, self._max_rmax = max_ring, max_rmax
        if drift_margin is None:
            # metres an owned floe may move between two box gathers (half of it, strictly): half the largest floe radius keeps the
            # extra halo small against the 2 x rmax interaction range (+ 25 % of a strip that holds ~6 % of a tile's floes at 8 tiles
            # of the 100 k field) and the gathers -- a handful of host synchronisations each -- a hundred steps apart at metres per step
            drift_margin = max(2000.0, 0.5 * max_rmax)
        w._chk(w.L.sz_tile_enable(w.h, capi.ptr(g, capi._lp), max_ring, max_rmax))
        self.world = w
        self.REC = w.L.sz_halo_record_doubles_ctx(w.h)          # (follows the largest ring of all ranks: 12 + 2 x ring capacity)
        self.dev = torch.device("cuda", device)
        self.cap = 0                                          # record slots per peer, sized from a counting pass
        self.send = self.recv = None
        if backend == "library":
            # Every step below that can fail on ONE rank is agreed on before the next collective: a rank that raised while its
            # peers waited in a broadcast or inside ncclCommInitRank would hang the job (no JSON line, no error message).
            uid = (C.c_char * 128)()
            if world > 1:
                self._agree(w.L.sz_comm_available() == 0, "RCCL cannot be bound by libsubzero_hip.so on some rank (librccl missing, or SZ_RCCL_DISABLE)")
                ok = True
                if rank == 0:
                    ok = w.L.sz_comm_unique_id(C.cast(uid, C.c_void_p)) == 0
                box = [bytes(uid) if ok else None]
                dist.broadcast_object_list(box, src=0)          # any host channel does: the id is 128 opaque bytes (None: rank 0 could not make one)
                if box[0] is None:
                    raise TileSetupError("ncclGetUniqueId failed on rank 0")
                uid = (C.c_char * 128).from_buffer_copy(box[0])
            rc = w.L.sz_comm_init(w.h, world, rank, C.cast(uid, C.c_void_p))
            self._agree(rc == 0, f"sz_comm_init failed on some rank ({w.L.sz_last_error(w.h).decode() if rc else 'not this one'})")
        elif backend == "library-host":
            self._transport = host_transport(dist, rank, world)          # (kept alive: the library calls into it)
            w._chk(w.L.sz_comm_init_host(w.h, world, rank, C.byref(self._transport)))
        if backend in ("library", "library-host"):
            w._chk(w.L.sz_tile_setup(w.h, self.L, self.L, int(self.per_x), int(self.per_y), float(drift_margin), int(rebox_every)))
            px, py = tile_grid(world)          # (the tile rule of assign_tiles: rank = iy * px + ix)
            w._chk(w.L.sz_tile_set_center(w.h, (rank % px + 0.5) * self.L / px, (rank // px + 0.5) * self.L / py))
        elif not host_staging:
            # kernels and RCCL collectives are ordered by ONE stream: no host sync inside a step
            w._chk(w.L.sz_set_stream(w.h, C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))
        self.margin, self.rebox_every = drift_margin, min(abs(rebox_every), 8)
        self._dt = 0
        self.rebox_every_max = abs(rebox_every)
        self._rebox_arg = rebox_every
        self._ref = None                                      # owned centroids at the last box gather (drift check)
        self.repartition_every = 500                          # steps between ownership checks (run())
        self.repartition_fraction = 0.1                       # ... re-tile when this share of the floes has left its tile
        self._since_check = 0
        self.boxes = None
        self.steps_since_box = 0
        self.tw_buf = None                                    # per-cell partial sums of the two-way coupling

    def _agree(self, ok, what):
        """collective: raise TileSetupError on every rank when `ok` is false on any of them"""
        if self.dist is None or self.nranks == 1:
            if not ok:
                raise TileSetupError(what)
            return
        t = self.torch.tensor([0 if ok else 1], dtype=self.torch.int32, device="cpu" if self.host_staging else self.torch.device("cuda", self.world_device))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        if int(t.item()):
            raise TileSetupError(what)

    # ---- collectives
    def _drift_since_last_gather(self):
        """largest centroid displacement of an owned floe since the last box gather, over all ranks (collective).  The
        halo selection is only valid while 2 x that stays under the drift margin built into the boxes: beyond it a
        contact across a tile edge could be missed silently, so it is an error; the gather interval adapts to the
        measured speed (40 % of the margin)."""
        w = self.world
        w._host_stale = True
        n = len(self.gidx)
        cx, cy = w.get("cx")[:n].copy(), w.get("cy")[:n].copy()
        d = 0.0
        if self._ref is not None and len(self._ref[0]) == n and n:
            dx, dy = np.abs(cx - self._ref[0]), np.abs(cy - self._ref[1])
            if self.per_x: dx = np.minimum(dx, np.abs(dx - self.L))       # a parent the ghost pass wrapped around
            if self.per_y: dy = np.minimum(dy, np.abs(dy - self.L))
            d = float(max(dx.max(), dy.max()))
        vmax = float(max(np.abs(w.get("u")[:n]).max(), np.abs(w.get("v")[:n]).max())) if n else 0.0
        t = self.torch.tensor([d, vmax], dtype=self.torch.float64, device="cpu" if self.host_staging else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        d, vmax = float(t[0].item()), float(t[1].item())
        if 2.0 * d > self.margin:
            raise capi.SzError(f"halo drift: a floe moved {d:.1f} m in {self.steps_since_box} steps, the margin is {self.margin:.1f} m "
                               f"(gather the boxes more often or widen the margin)")
        # the faster of the measured displacement per step and the largest velocity component now may use 30 % of the margin before the
        # next gather; the interval at most doubles from one gather to the next (floes that start from rest speed up)
        per_step = max(d / self.steps_since_box if self.steps_since_box > 0 else 0.0, vmax * abs(self._dt))
        if per_step > 0:
            self.rebox_every = int(max(1, min(self.rebox_every_max, 2 * self.rebox_every if self.steps_since_box > 0 else self.rebox_every,
                                              0.3 * self.margin / per_step)))
        self._ref = (cx, cy)

    def _allgather_boxes(self):
        w = self.world
        b5 = np.zeros(5)
        w._chk(w.L.sz_sync(w.h))
        self._drift_since_last_gather()
        w._chk(w.L.sz_owned_box(w.h, capi.ptr(b5)))
        t = self.torch.tensor(b5, dtype=self.torch.float64, device="cpu" if self.host_staging else self.dev)
        out = [self.torch.zeros_like(t) for _ in range(self.nranks)]
        self.dist.all_gather(out, t)
        allb = np.stack([o.cpu().numpy() for o in out])
        allb[:, 4] = allb[:, 4].max()
        self.boxes = np.ascontiguousarray(np.stack([expanded_box(b, self.margin) for b in allb]))
        w._chk(w.L.sz_halo_set_boxes(w.h, self.nranks, capi.ptr(self.boxes)))
        self.steps_since_box = 0
        # size the exchange buffers: counting pass, largest count over all ranks and peers, 50 % head room
        # (the all-to-all uses equal splits, so every rank must use the same number of slots)
        w._chk(w.L.sz_halo_pack(w.h, self.nranks, self.rank, self.L, self.L, int(self.per_x), int(self.per_y), None, 1))
        counts = np.zeros(self.nranks, np.int32)
        w._chk(w.L.sz_halo_counts(w.h, self.nranks, capi.ptr(counts, capi._ip)))
        t = self.torch.tensor([int(counts.max())], dtype=self.torch.int64, device="cpu" if self.host_staging else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        cap = int(t.item()) * 3 // 2 + 32
        if cap > self.cap:
            self.cap = cap
            n = self.nranks * (self.cap + 1) * self.REC
            self.send = self.torch.zeros(n, dtype=self.torch.float64, device=self.dev)
            self.recv = self.torch.zeros(n, dtype=self.torch.float64, device=self.dev)

    def exchange(self, between=None):
        """steps 1-2 of the module docstring, asynchronous on the device: fills self.recv.  `between` is enqueued
        after the pack kernel and before the wait for the collective: work that does not need the halo (the
        forcings of the owned floes) then runs beside the exchange."""
        torch, dist, w = self.torch, self.dist, self.world
        if self.boxes is None or self.steps_since_box >= self.rebox_every:
            self._allgather_boxes()
        self.steps_since_box += 1
        w._chk(w.L.sz_halo_pack(w.h, self.nranks, self.rank, self.L, self.L, int(self.per_x), int(self.per_y),
                                C.c_void_p(self.send.data_ptr()), self.cap))
        # (this path sizes every region alike and has no per-step drift check: after a box gather the floes may move
        #  margin / 2 -- _allgather_boxes measures what they really moved and shortens the interval when needed)
        if self.host_staging:                       # gloo: through the host
            if between:
                between()
            w._chk(w.L.sz_sync(w.h))
            sb = self.send.cpu(); rb = torch.zeros_like(sb)
            dist.all_to_all_single(rb, sb)
            self.recv.copy_(rb)
            torch.cuda.synchronize()
        else:                                       # RCCL: equal splits, device to device
            work = dist.all_to_all_single(self.recv, self.send, async_op=True)
            if between:
                between()
            work.wait()                             # stream-level: the kernels enqueued next wait for the collective

    def step(self, tstep, dt, coupling_dt=10, collisions_on=True, coupling_on=True):
        w = self.world
        self._dt = dt
        flags = (capi.COLLISIONS_ON if collisions_on else 0) | (capi.COUPLING_ON if coupling_on else 0)
        peers = self.nranks > 1 or self.always_exchange
        if peers:
            self.exchange(lambda: w._chk(w.L.sz_tile_forcing(w.h, int(tstep), int(coupling_dt), flags)))
        w._chk(w.L.sz_tile_step(w.h, C.c_void_p(self.recv.data_ptr()) if peers else None, self.nranks if peers else 0,
                                self.cap, int(tstep), int(dt), int(coupling_dt), flags))
        if self.tw_buf is not None and coupling_on and coupling_dt > 0 and tstep % coupling_dt == 0:
            self._two_way_reduce(dt)
        w._host_stale = True

    # ---- two-way coupling across tiles
    def set_two_way(self, t_ocn, t_atm, dt, Cd_ao=1.25e-3, k=2.14, L=2.93e5):
        """calc_two_way_coupling! (coupling.jl:1617-1680) in a tiled run: every rank sums the ice stress per centre
        cell over the floes it owns, the partial fields are added up with one all-reduce per coupling step, and
        every rank finishes the cells (so every rank holds the full ocean fields, like the single-process run)."""
        w = self.world
        w.set_two_way(True, Cd_ao=Cd_ao, k=k, L=L, dt=dt); w.set_temps(t_ocn, t_atm)
        Nx, Ny = w._grid
        self.tw_buf = self.torch.zeros(3 * (Nx + 1) * (Ny + 1), dtype=self.torch.float64, device=self.dev)

    def _two_way_reduce(self, dt):
        w, torch = self.world, self.torch
        w._chk(w.L.sz_two_way_partial(w.h, C.c_void_p(self.tw_buf.data_ptr())))
        if self.nranks > 1:
            if self.host_staging:
                w._chk(w.L.sz_sync(w.h))
                hb = self.tw_buf.cpu(); self.dist.all_reduce(hb); self.tw_buf.copy_(hb); torch.cuda.synchronize()
            else:
                self.dist.all_reduce(self.tw_buf)
        w._chk(w.L.sz_two_way_finish(w.h, C.c_void_p(self.tw_buf.data_ptr()), int(dt)))

    # ---- grid output across tiles
    def write_grid_data(self, xg, yg, outputs=None):
        """write_grid_data! (output.jl:588-605) in a tiled run: every rank sums what its own floes and their ghosts
        put into each output cell, one all-reduce adds the partial fields up, every rank finishes the averages
        (array [len(outputs), nx, ny] on every rank, as World.write_grid_data gives it)."""
        w, torch = self.world, self.torch
        names = list(w.EUL_OUTPUTS if outputs is None else outputs)
        codes = np.array([w.EUL_OUTPUTS.index(n) for n in names], np.int32)
        xg = np.ascontiguousarray(xg, np.float64); yg = np.ascontiguousarray(yg, np.float64)
        nx, ny = len(xg) - 1, len(yg) - 1
        buf = torch.zeros(capi.EUL_PARTIAL * nx * ny, dtype=torch.float64, device=self.dev)
        self.sync()
        w.add_ghosts()
        try:
            w._chk(w.L.sz_eulerian_partial(w.h, nx, ny, capi.ptr(xg), capi.ptr(yg), C.c_void_p(buf.data_ptr())))
        finally:
            w.remove_ghosts()
        if self.nranks > 1:
            if self.host_staging:
                hb = buf.cpu(); self.dist.all_reduce(hb); buf.copy_(hb)
            else:
                self.dist.all_reduce(buf)
            torch.cuda.synchronize()
        out = np.zeros((max(len(names), 1), nx, ny))
        w._chk(w.L.sz_eulerian_finish(w.h, nx, ny, capi.ptr(xg), capi.ptr(yg), C.c_void_p(buf.data_ptr()), len(names),
                                      capi.ptr(codes, capi._ip), capi.ptr(out)))
        return out[:len(names)]

    # ---- migration: rare, host-side
    def repartition(self, owner_fn=None):
        """Re-assign every floe to the tile its centroid lies in now and move the floes that changed tile to
        their new owner with their complete state (all columns incl. the previous-step tendencies, tensors,
        status, ring, sub-floe points), so that the run continues exactly as it would have.  Floes move metres per
        step against tiles of hundreds of kilometres: this is a rare operation, done on the host (download,
        all_gather_object of the movers, upload).  Returns the number of floes this rank gave away.
        owner_fn(cx, cy) -> rank array overrides the tile rule (tests)."""
        w, dist, rank = self.world, self.dist, self.rank
        self.sync(); w._host_stale = True; w._pull()
        col = w.col; n = len(self.gidx)
        assert w._M == n, "repartition needs a ghost-free state"
        cx, cy = col["cx"].copy(), col["cy"].copy()
        if self.per_x: cx %= self.L
        if self.per_y: cy %= self.L
        owner = np.asarray(owner_fn(cx, cy) if owner_fn else assign_tiles(cx, cy, self.L, self.nranks))
        mv = np.nonzero(owner != rank)[0]; keep = np.nonzero(owner == rank)[0]

        def pick(idx):
            """the rows `idx` of the local columns as one record batch"""
            off = col["vert_off"]; so = col["sub_off"]
            rec = {"gidx": self.gidx[idx], "owner": owner[idx]}
            for k in capi.DCOLS + capi.TCOLS + ["id", "status"]:
                rec[k] = col[k][idx]
            rec["nv"] = np.diff(off)[idx]; rec["ns"] = np.diff(so)[idx]
            sel = np.concatenate([np.arange(off[i], off[i + 1]) for i in idx]) if len(idx) else np.zeros(0, int)
            ssel = np.concatenate([np.arange(so[i], so[i + 1]) for i in idx]) if len(idx) else np.zeros(0, int)
            rec["vx"], rec["vy"] = col["vx"][sel], col["vy"][sel]
            rec["sx"], rec["sy"] = col["sx"][ssel], col["sy"][ssel]
            return rec

        out = pick(mv)
        gathered = [None] * self.nranks
        dist.all_gather_object(gathered, out)
        parts = [pick(keep)]
        for r, rec in enumerate(gathered):
            if r == rank or len(rec["gidx"]) == 0:
                continue
            m = np.nonzero(rec["owner"] == rank)[0]
            if len(m) == 0:
                continue
            vo = np.concatenate([[0], np.cumsum(rec["nv"])]); so = np.concatenate([[0], np.cumsum(rec["ns"])])
            sub = {k: rec[k][m] for k in rec if k not in ("vx", "vy", "sx", "sy")}
            vs = np.concatenate([np.arange(vo[i], vo[i + 1]) for i in m]); ss = np.concatenate([np.arange(so[i], so[i + 1]) for i in m])
            sub["vx"], sub["vy"], sub["sx"], sub["sy"] = rec["vx"][vs], rec["vy"][vs], rec["sx"][ss], rec["sy"][ss]
            parts.append(sub)
        total_moved = int(sum(len(g["gidx"]) for g in gathered))
        if total_moved == 0:
            return 0
        # merge, ordered by global index (any fixed order would do: order-dependent rules use the global index)
        g_all = np.concatenate([p_["gidx"] for p_ in parts]); order = np.argsort(g_all, kind="stable")
        nv_all = np.concatenate([p_["nv"] for p_ in parts]); ns_all = np.concatenate([p_["ns"] for p_ in parts])
        vstart = np.concatenate([[0], np.cumsum(nv_all)]); sstart = np.concatenate([[0], np.cumsum(ns_all)])
        vx_all = np.concatenate([p_["vx"] for p_ in parts]); vy_all = np.concatenate([p_["vy"] for p_ in parts])
        sx_all = np.concatenate([p_["sx"] for p_ in parts]); sy_all = np.concatenate([p_["sy"] for p_ in parts])
        cols = {k: np.concatenate([p_[k] for p_ in parts])[order] for k in capi.DCOLS + capi.TCOLS + ["id", "status"]}
        nv = nv_all[order]; ns = ns_all[order]
        cols["vert_off"] = np.concatenate([[0], np.cumsum(nv)]).astype(np.int32)
        vsel = np.concatenate([np.arange(vstart[i], vstart[i + 1]) for i in order]) if len(order) else np.zeros(0, int)
        ssel = np.concatenate([np.arange(sstart[i], sstart[i + 1]) for i in order]) if len(order) else np.zeros(0, int)
        cols["vx"], cols["vy"] = vx_all[vsel], vy_all[vsel]
        soff = np.concatenate([[0], np.cumsum(ns)]).astype(np.int32)
        self.gidx = g_all[order]
        w.load_columns(cols); w.set_subpoints_csr(soff, sx_all[ssel], sy_all[ssel])
        w._push()
        g = np.ascontiguousarray(self.gidx, np.int64)
        w._chk(w.L.sz_tile_enable(w.h, capi.ptr(g, capi._lp), self._max_ring, self._max_rmax))
        if self.backend in ("library", "library-host"):
            w._chk(w.L.sz_tile_setup(w.h, self.L, self.L, int(self.per_x), int(self.per_y), float(self.margin), int(self._rebox_arg)))
        self.boxes = None                      # owned boxes and halo capacity are re-established at the next exchange
        self._ref = None
        return len(mv)

    def migrate(self, owner_fn=None):
        """Migration inside the library (sz_tile_migrate: library backends): every owned floe goes to the tile that holds its centroid now
        (or to owner_fn(cx, cy)), with its complete state, over the library's own channel (RCCL / the host transport); the context is
        rebuilt from the kept and received floes ordered by global index.  Collective.  Returns the number of floes this rank gave away."""
        w = self.world
        self.sync()
        ov = None
        if owner_fn is not None:
            w._host_stale = True
            n = len(self.gidx)
            cx, cy = w.get("cx")[:n].copy(), w.get("cy")[:n].copy()
            if self.per_x: cx %= self.L
            if self.per_y: cy %= self.L
            ov = np.ascontiguousarray(owner_fn(cx, cy), np.int32)
        px, py = tile_grid(self.nranks)
        sent = C.c_int64(0); owned = C.c_int64(0)
        w._chk(w.L.sz_tile_migrate(w.h, int(px), int(py), capi.ptr(ov, capi._ip) if ov is not None else None, C.byref(sent), C.byref(owned)))
        # the host's copies of the columns are stale now: the columns come back with the next get() (World._pull), the sub-floe points when
        # something asks for them (World._fetch_subpoints) -- a migration itself moves no floe data to the host
        n = int(owned.value)
        w.N = n; w._M = n; w._dirty = False; w._host_stale = True; w._sub = {}; w._sub_on_device = True
        g = np.zeros(max(n, 1), np.int64)
        w._chk(w.L.sz_tile_owned_gidx(w.h, capi.ptr(g, capi._lp), int(g.size)))
        self.gidx = g[:n].copy()
        self.migrate_path = int(w.L.sz_debug_migrate_path(w.h))          # 1: packed on the device, 2: host-staged
        self.boxes = None; self._ref = None
        return int(sent.value)

    def maybe_repartition(self):
        """Ownership is static between calls of repartition(); floes drift.  Collective: re-tile when more than
        `repartition_fraction` of all floes has left the tile that owns it (cheap test: the owned centroids against the
        tile rule).  Returns the number of floes this rank gave away (0: nothing done)."""
        self.sync(); w = self.world; w._host_stale = True
        n = len(self.gidx)
        cx, cy = w.get("cx")[:n].copy(), w.get("cy")[:n].copy()
        if self.per_x: cx %= self.L
        if self.per_y: cy %= self.L
        away = int(np.count_nonzero(assign_tiles(cx, cy, self.L, self.nranks) != self.rank)) if n else 0
        t = self.torch.tensor([away, n], dtype=self.torch.int64, device="cpu" if self.host_staging else self.dev)
        self.dist.all_reduce(t)
        if int(t[0]) <= self.repartition_fraction * max(int(t[1]), 1):
            return 0
        return self.migrate() if self.backend in ("library", "library-host") else self.repartition()

    def run(self, nsteps, tstep0, dt, coupling_dt=10, collisions_on=True, coupling_on=True, stop_on_tags=False):
        """nsteps x timestep_sim! of the tiled run (collective).  Returns the steps run.  stop_on_tags (library backends): the batch ends
        after the first step that leaves a floe tagged remove / fuse on ANY rank -- the same step on every rank -- as World.run does
        for the single context (the reference runs simplify_floes! after every step, simulation.jl:205-214)."""
        done = 0
        while done < nsteps:
            k = min(nsteps - done, self.repartition_every - self._since_check) if self.nranks > 1 else nsteps - done
            if self.backend in ("library", "library-host"):
                w = self.world
                flags = (capi.COLLISIONS_ON if collisions_on else 0) | (capi.COUPLING_ON if coupling_on else 0) | (0 if stop_on_tags else capi.NO_STOP)
                ran = C.c_int32(0)
                w._chk(w.L.sz_tile_run(w.h, int(k), int(tstep0 + done), int(dt), int(coupling_dt), flags, C.byref(ran)))
                w._host_stale = True
                if ran.value < k:
                    return done + int(ran.value)
            else:
                for s in range(k):
                    self.step(tstep0 + done + s, dt, coupling_dt, collisions_on, coupling_on)
                self.sync()
            done += k; self._since_check += k
            if self.nranks > 1 and self._since_check >= self.repartition_every:
                self._since_check = 0
                self.maybe_repartition()
        return done

    def sync(self):
        self.world._chk(self.world.L.sz_sync(self.world.h))

    @property
    def n_halo_last(self):
        self.sync()
        return self.world.stats().get("n_halo", 0)

    def owned(self, name):
        """column `name` of the owned floes (global indices in self.gidx)."""
        return self.world.get(name)[:len(self.gidx)]
