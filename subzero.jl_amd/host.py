"""Host-side mirror of the reference's process API for the hot path, on top of the C-ABI.

`World` keeps the hot columns of the reference's StructArray{Floe} as numpy arrays, hands them to
libsubzero_hip.so, and exposes one method per reference function (same names, argument meaning
and mutation contract):

    add_ghosts!                  -> World.add_ghosts()                 collisions.jl:1060
    timestep_collisions!         -> World.timestep_collisions(n_init, dt)          :734
    floe_floe_interaction!       -> World.floe_floe_interaction(i, j, dt, max_overlap)  :347
    floe_domain_interaction!     -> World.floe_domain_interaction(i, dt, max_overlap)   :594
    timestep_coupling!           -> World.timestep_coupling()          coupling.jl:1705
    timestep_floe_properties!    -> World.timestep_floe_properties(dt) update_floe.jl:469
    timestep_sim!                -> World.timestep_sim(tstep, dt, ...) / World.run(nsteps, ...)  simulation.jl:94

All arithmetic of those calls happens in HIP kernels; this module only packs and unpacks
columns.  Without the built library and a HIP device every call raises (no CPU fallback).
"""
import ctypes as C

import numpy as np

from . import capi, floe as floe_mod
from .capi import SzError

_I32 = np.int32
_I64 = np.int64


class World:
    def __init__(self, device=0, ftype=np.float64):
        """ftype = numpy.float32: a Floe{Float32} host -- the columns cross the boundary as floats (sz_*_f32); the engine computes in double
        all the same, and this class keeps its own copies as float64 holding the rounded values"""
        self._ft = np.dtype(ftype)
        assert self._ft in (np.dtype(np.float64), np.dtype(np.float32))
        self.L = capi.load()
        self.h = self.L.sz_create(int(device))
        if not self.h:
            raise SzError("sz_create failed: no HIP device available (the HIP engine has no CPU fallback)")
        self.h = C.c_void_p(self.h)
        self._params = dict(E=6e6, nu=0.3, mu=0.2, rho_o=1027.0, rho_a=1.2, Cd_io=3e-3, Cd_ia=1e-3, f=1.4e-4,
                            turn_theta=15 * np.pi / 180, floe_floe_max_overlap=0.55, floe_domain_max_overlap=0.75,
                            rho_i=920.0, max_floe_height=10.0, maximum_xi=1e-5, lam=0.2, coupling_dd=1)
        self._rings = []          # host rings of floes added with add_floe
        self._sub = {}            # i -> (sx, sy)
        self.col = {}             # host columns (valid when not _host_stale)
        self.N = 0
        self._M = 0
        self._dirty = True        # host columns changed since the last upload
        self._host_stale = False  # device columns changed since the last download
        self._inter = {}          # hand-set interaction matrices waiting for upload
        self._have_domain = False
        self._domain = None
        self._topo = []

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.sz_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _chk(self, rc):
        if rc != 0:
            raise SzError(f"libsubzero_hip error {rc}: {self.L.sz_last_error(self.h).decode()}")

    def _push_params(self):
        p = capi.SzParams(**self._params, _pad=0)
        self._chk(self.L.sz_set_params(self.h, C.byref(p)))

    # ------------------------------------------------------------------ static inputs
    def set_consts(self, E=6e6, nu=0.3, mu=0.2, rho_o=1027.0, rho_a=1.2, Cd_io=3e-3, Cd_ia=1e-3, f=1.4e-4,
                   turn_theta=15 * np.pi / 180):
        self._params.update(E=E, nu=nu, mu=mu, rho_o=rho_o, rho_a=rho_a, Cd_io=Cd_io, Cd_ia=Cd_ia, f=f,
                            turn_theta=turn_theta)
        self._push_params()

    def set_settings(self, floe_floe_max_overlap=0.55, floe_domain_max_overlap=0.75, rho_i=920.0,
                     max_floe_height=10.0, maximum_xi=1e-5, lam=0.2, coupling_dd=1):
        self._params.update(floe_floe_max_overlap=floe_floe_max_overlap, floe_domain_max_overlap=floe_domain_max_overlap,
                            rho_i=rho_i, max_floe_height=max_floe_height, maximum_xi=maximum_xi, lam=lam,
                            coupling_dd=int(coupling_dd))
        self._push_params()

    def set_domain(self, kinds, x0, xf, y0, yf, bu=None, bv=None):
        rects, vals = floe_mod.boundary_rects(x0, xf, y0, yf)
        self.set_domain_raw(kinds, vals, rects, bu, bv)
        self._extent = (x0, xf, y0, yf)

    def set_domain_raw(self, kinds, vals, rects, bu=None, bv=None):
        k = np.ascontiguousarray(kinds, _I32)
        vals = np.ascontiguousarray(vals, np.float64); rects = np.ascontiguousarray(rects, np.float64)
        bu = np.ascontiguousarray(bu if bu is not None else np.zeros(4), np.float64)
        bv = np.ascontiguousarray(bv if bv is not None else np.zeros(4), np.float64)
        self._chk(self.L.sz_set_domain(self.h, capi.ptr(k, capi._ip), capi.ptr(vals), capi.ptr(rects), capi.ptr(bu), capi.ptr(bv)))
        self._have_domain = True
        self._domain = (k, vals, rects, bu, bv)
        self._push_params()

    def set_topography(self, rings):
        rings = [floe_mod.valid_ring(r) for r in rings]
        off = np.zeros(len(rings) + 1, _I32)
        props = []
        for i, r in enumerate(rings):
            off[i + 1] = off[i] + len(r)
            props.append(floe_mod.topography_props(r))
        xy = np.concatenate(rings, 0) if rings else np.zeros((0, 2))
        x = np.ascontiguousarray(xy[:, 0]); y = np.ascontiguousarray(xy[:, 1])
        cx = np.array([p[0] for p in props], np.float64); cy = np.array([p[1] for p in props], np.float64)
        rm = np.array([p[2] for p in props], np.float64)
        self._chk(self.L.sz_set_topography(self.h, len(rings), capi.ptr(off, capi._ip), capi.ptr(x), capi.ptr(y),
                                           capi.ptr(cx), capi.ptr(cy), capi.ptr(rm)))

    def set_grid_fields(self, Nx, Ny, x0, xf, y0, yf, uo, vo, hflx, ua, va):
        arrs = [np.ascontiguousarray(np.broadcast_to(a, (Nx + 1, Ny + 1)), self._ft) for a in (uo, vo, hflx, ua, va)]
        f32 = self._ft == np.float32
        fn = self.L.sz_set_fields_f32 if f32 else self.L.sz_set_fields
        self._chk(fn(self.h, int(Nx), int(Ny), float(x0), float(xf), float(y0), float(yf), *(capi.ptr(a, capi._fp if f32 else capi._dp) for a in arrs)))
        self._grid = (int(Nx), int(Ny))

    def set_precision(self, mode):
        """"f64" (default) or "mixed": the per-point arithmetic of the forcings in fp32 (BASELINE configs[4])"""
        self._chk(self.L.sz_set_precision(self.h, {"f64": 0, "mixed": 1}[mode]))

    # ---- two-way coupling (coupling.jl:1617-1680); off by default like CouplingSettings()
    def set_two_way(self, on=True, Cd_ao=1.25e-3, k=2.14, L=2.93e5, dt=10):
        self._chk(self.L.sz_set_two_way(self.h, int(on), float(Cd_ao), float(k), float(L), int(dt)))

    def set_temps(self, t_ocn, t_atm):
        Nx, Ny = self._grid
        a, b = (np.ascontiguousarray(np.broadcast_to(t, (Nx + 1, Ny + 1)), np.float64) for t in (t_ocn, t_atm))
        self._chk(self.L.sz_set_temps(self.h, capi.ptr(a), capi.ptr(b)))

    def ocean_stress(self):
        """tau_x, tau_y, si_frac, hflx_factor on the (Nx+1) x (Ny+1) grid-line lattice"""
        Nx, Ny = self._grid
        out = [np.zeros((Nx + 1, Ny + 1)) for _ in range(4)]
        self._chk(self.L.sz_download_ocean_stress(self.h, *(capi.ptr(a) for a in out)))
        return out

    # ------------------------------------------------------------------ floe setup (host side)
    def add_floe(self, coords, height):
        """Floe(coords, hmean, 0): appends a parent floe; derived columns as the reference computes them."""
        self._pull()
        ring = floe_mod.valid_ring(coords)
        d = floe_mod.derive(np.array([0, len(ring)]), ring[:, 0].copy(), ring[:, 1].copy(), height,
                            self._params["rho_i"])
        i = self.N
        if not self.col:
            self.col = {n: np.zeros(0) for n in capi.DCOLS}
            for n in capi.TCOLS:
                self.col[n] = np.zeros((0, 4))
            self.col["id"] = np.zeros(0, _I64); self.col["ghost_id"] = np.zeros(0, _I64)
            self.col["status"] = np.zeros(0, _I32)
            self.col["vert_off"] = np.zeros(1, _I32); self.col["vx"] = np.zeros(0); self.col["vy"] = np.zeros(0)
        assert self._M == self.N, "add_floe while ghosts exist"
        for n in capi.DCOLS:
            self.col[n] = np.append(self.col[n], d[n][0] if n in d else 0.0)
        for n in capi.TCOLS:
            self.col[n] = np.vstack([self.col[n], np.zeros((1, 4))])
        self.col["id"] = np.append(self.col["id"], _I64(i + 1))
        self.col["ghost_id"] = np.append(self.col["ghost_id"], _I64(0))
        self.col["status"] = np.append(self.col["status"], _I32(capi.ACTIVE))
        self.col["vert_off"] = np.append(self.col["vert_off"], _I32(self.col["vert_off"][-1] + len(ring)))
        self.col["vx"] = np.append(self.col["vx"], ring[:, 0]); self.col["vy"] = np.append(self.col["vy"], ring[:, 1])
        self.N += 1; self._M += 1
        self._dirty = True
        return i

    def load_columns(self, cols, N=None):
        """Bulk initialisation from ready-made columns (dict with the sz_floe_columns names)."""
        self.col = {k: np.ascontiguousarray(v) for k, v in cols.items()}
        M = len(self.col["cx"])
        self.N = M if N is None else int(N); self._M = M
        for n in capi.DCOLS:
            self.col.setdefault(n, np.zeros(M))
        for n in capi.TCOLS:
            self.col.setdefault(n, np.zeros((M, 4)))
        self.col.setdefault("id", np.arange(1, M + 1, dtype=_I64))
        self.col.setdefault("ghost_id", np.zeros(M, _I64))
        self.col.setdefault("status", np.full(M, capi.ACTIVE, _I32))
        self._sub = {}; self._sub_on_device = False
        self._dirty = True; self._host_stale = False
        self._new_field = True    # a different field: the interaction rows the device may still hold are not its rows

    def set_subpoints(self, i, sx, sy):
        self._sub[int(i)] = (np.ascontiguousarray(sx, np.float64), np.ascontiguousarray(sy, np.float64))
        self._dirty = True

    def set_subpoints_csr(self, sub_off, sx, sy):
        self._sub_on_device = False
        self.col["sub_off"] = np.ascontiguousarray(sub_off, _I32)
        self.col["sx"] = np.ascontiguousarray(sx, np.float64); self.col["sy"] = np.ascontiguousarray(sy, np.float64)
        self._dirty = True

    # ------------------------------------------------------------------ host <-> device
    def _columns_struct(self, col, keep):
        f32 = self._ft == np.float32
        f = capi.SzFloeColumnsF32() if f32 else capi.SzFloeColumns()
        pt = capi._fp if f32 else capi._dp
        for n in capi.DCOLS + ["vx", "vy", "sx", "sy"]:
            a = col.get(n)
            if a is not None:
                a = np.ascontiguousarray(a, self._ft); keep.append(a); setattr(f, n, capi.ptr(a, pt))
        for n in capi.TCOLS:
            a = col.get(n)
            if a is not None:
                a = np.ascontiguousarray(a, self._ft).reshape(-1); keep.append(a); setattr(f, n, capi.ptr(a, pt))
        for n in ("id", "ghost_id"):
            a = col.get(n)
            if a is not None:
                a = np.ascontiguousarray(a, _I64); keep.append(a); setattr(f, n, capi.ptr(a, capi._lp))
        for n in ("status", "vert_off", "sub_off", "ghost_off", "ghost_idx"):
            a = col.get(n)
            if a is not None:
                a = np.ascontiguousarray(a, _I32); keep.append(a); setattr(f, n, capi.ptr(a, capi._ip))
        return f

    def subpoints(self):
        """(offsets, sx, sy) of the sub-floe points as the library holds them (sz_download_subpoints): the caller's points in the caller's order"""
        self._push()
        n = self.N
        off = np.zeros(n + 1, _I32)
        self._chk(self.L.sz_download_subpoints(self.h, capi.ptr(off, capi._ip), None, None))
        sx = np.zeros(max(int(off[n]), 1)); sy = np.zeros(max(int(off[n]), 1))
        self._chk(self.L.sz_download_subpoints(self.h, capi.ptr(off, capi._ip), capi.ptr(sx), capi.ptr(sy)))
        return off, sx[:off[n]], sy[:off[n]]

    def _fetch_subpoints(self):
        """after a migration of a tiled run (tiles.TiledWorld.migrate) the sub-floe points of the tile are the library's: fetched when needed"""
        if not getattr(self, "_sub_on_device", False):
            return
        n = self.N
        off = np.zeros(n + 1, _I32)
        self._chk(self.L.sz_download_subpoints(self.h, capi.ptr(off, capi._ip), None, None))
        sx = np.zeros(max(int(off[n]), 1)); sy = np.zeros(max(int(off[n]), 1))
        self._chk(self.L.sz_download_subpoints(self.h, capi.ptr(off, capi._ip), capi.ptr(sx), capi.ptr(sy)))
        self.col["sub_off"], self.col["sx"], self.col["sy"] = off, sx[:off[n]], sy[:off[n]]
        self._sub_on_device = False

    def _push(self):
        if not self._dirty:
            return
        if not self._have_domain:
            raise SzError("set_domain must be called before the first process call")
        self._fetch_subpoints()
        col = dict(self.col)
        if "sub_off" not in col or self._sub:
            off = np.zeros(self.N + 1, _I32); xs = []; ys = []
            for i in range(self.N):
                sx, sy = self._sub.get(i, (np.zeros(0), np.zeros(0)))
                if "sub_off" in self.col and i not in self._sub:
                    o0, o1 = self.col["sub_off"][i], self.col["sub_off"][i + 1]
                    sx, sy = self.col["sx"][o0:o1], self.col["sy"][o0:o1]
                off[i + 1] = off[i] + len(sx); xs.append(sx); ys.append(sy)
            col["sub_off"] = off
            col["sx"] = np.concatenate(xs) if xs else np.zeros(0)
            col["sy"] = np.concatenate(ys) if ys else np.zeros(0)
            self.col["sub_off"], self.col["sx"], self.col["sy"] = col["sub_off"], col["sx"], col["sy"]
            self._sub = {}
        keep = []
        f = self._columns_struct(col, keep)
        self._chk((self.L.sz_upload_floes_f32 if self._ft == np.float32 else self.L.sz_upload_floes)(self.h, int(self._M), int(self.N), C.byref(f)))
        self._dirty = False; self._host_stale = False
        if getattr(self, "_new_field", False):
            off = np.zeros(self._M + 1, _I32)
            self._chk(self.L.sz_upload_interactions(self.h, capi.ptr(off, capi._ip), None))
            self._new_field = False

    def crec_mismatches(self):
        """test hook: parts of the collision records (the per-floe cache of the columns the neighbour search and the narrow phase read) that differ
        from the columns after the last resident batch; -1: that batch did not run on records"""
        n = C.c_int64(0)
        self._chk(self.L.sz_debug_crec_mismatches(self.h, C.byref(n)))
        return int(n.value)

    def find_key(self, slot, key):
        """diagnosis: the ghost / halo row with order key `key` of the last resident step that used ghost allocator `slot` (sz_debug_find_key)"""
        out = np.zeros(56)
        self._chk(self.L.sz_debug_find_key(self.h, int(slot), int(key), capi.ptr(out)))
        n = int(out[13])
        return None if out[0] < 0 else {"row": int(out[0]), "cx": out[1], "cy": out[2], "u": out[3], "v": out[4], "xi": out[5], "rmax": out[6], "area": out[7],
                                        "height": out[8], "box": out[9:13].copy(), "parent": int(out[14]), "status": int(out[15]),
                                        "x": out[16:16 + n].copy(), "y": out[36:36 + n].copy()}

    def pairs_of_ids(self, slot, id_a, id_b):
        """diagnosis: (owner key, partner key, contact rows, owner row, partner row) of the last resident step's pair items between instances of two ids"""
        out = np.zeros(61)
        self._chk(self.L.sz_debug_pairs_of_ids(self.h, int(slot), int(id_a), int(id_b), capi.ptr(out)))
        return [tuple(int(v) for v in out[1 + 5 * k:6 + 5 * k]) for k in range(min(int(out[0]), 12))]

    def stats(self):
        s = capi.SzStats()
        self._chk(self.L.sz_get_stats(self.h, C.byref(s)))
        return {n: int(getattr(s, n)) for n, _ in capi.SzStats._fields_}

    def _pull(self):
        if not self._host_stale:
            return
        st = self.stats()
        M, V = st["M"], st["n_ring_points"]
        col = {n: np.zeros(M, self._ft) for n in capi.DCOLS}
        for n in capi.TCOLS:
            col[n] = np.zeros((M, 4), self._ft)
        col["id"] = np.zeros(M, _I64); col["ghost_id"] = np.zeros(M, _I64); col["status"] = np.zeros(M, _I32)
        col["vert_off"] = np.zeros(M + 1, _I32); col["vx"] = np.zeros(V, self._ft); col["vy"] = np.zeros(V, self._ft)
        col["ghost_off"] = np.zeros(M + 1, _I32); col["ghost_idx"] = np.zeros(max(3 * M, 1), _I32)
        keep = []
        f = self._columns_struct(col, keep)
        self._chk((self.L.sz_download_floes_f32 if self._ft == np.float32 else self.L.sz_download_floes)(self.h, C.byref(f)))
        if self._ft == np.float32:          # (this class keeps float64 copies: the rounded values)
            for n in list(col):
                if col[n].dtype == np.float32:
                    col[n] = col[n].astype(np.float64)
        self._fetch_subpoints()
        for n in ("sub_off", "sx", "sy"):
            if n in self.col:
                col[n] = self.col[n]
        col["ghost_idx"] = col["ghost_idx"][:col["ghost_off"][M]]
        self.col = col
        self._M = M
        self._host_stale = False

    # ------------------------------------------------------------------ state access (oracle-compatible)
    @property
    def M(self):
        self._pull()
        return self._M

    def get(self, name):
        self._pull()
        m = {"sa": "stress_accum", "si": "stress_instant", "e": "strain"}
        for pre, full in m.items():
            if name.startswith(pre) and name[len(pre):] in ("11", "12", "21", "22"):
                return self.col[full][:, ("11", "12", "21", "22").index(name[len(pre):])].copy()
        return self.col[name].copy()

    def set(self, name, vals):
        self._pull()
        v = np.ascontiguousarray(np.broadcast_to(vals, (self._M,)), np.float64).copy()
        for pre, full in {"sa": "stress_accum", "si": "stress_instant", "e": "strain"}.items():
            if name.startswith(pre) and name[len(pre):] in ("11", "12", "21", "22"):
                self.col[full][:, ("11", "12", "21", "22").index(name[len(pre):])] = v
                break
        else:
            self.col[name] = v
        self._dirty = True

    def ids(self):
        self._pull()
        return self.col["id"].copy(), self.col["ghost_id"].copy(), self.col["status"].copy()

    def set_ids(self, ids):
        self._pull(); self.col["id"] = np.ascontiguousarray(ids, _I64); self._dirty = True

    def set_status(self, st):
        self._pull(); self.col["status"] = np.ascontiguousarray(st, _I32); self._dirty = True

    def rings(self):
        self._pull()
        return self.col["vert_off"], self.col["vx"], self.col["vy"]

    def ring(self, i):
        off, x, y = self.rings()
        return np.stack([x[off[i]:off[i + 1]], y[off[i]:off[i + 1]]], 1)

    def interactions(self):
        self._push()
        st = self.stats()
        off = np.zeros(st["M"] + 1, _I32); rows = np.zeros((max(st["n_inter_rows"], 1), 7), self._ft)
        if self._ft == np.float32:
            self._chk(self.L.sz_download_interactions_f32(self.h, capi.ptr(off, capi._ip), capi.ptr(rows, capi._fp)))
            rows = rows.astype(np.float64)
        else:
            self._chk(self.L.sz_download_interactions(self.h, capi.ptr(off, capi._ip), capi.ptr(rows)))
        return off, rows[:off[-1]]

    def inter(self, i):
        off, rows = self.interactions()
        return rows[off[i]:off[i + 1]]

    def ghosts(self):
        self._pull()
        if "ghost_off" not in self.col:
            return [[] for _ in range(self._M)]
        o, g = self.col["ghost_off"], self.col["ghost_idx"]
        return [list(g[o[i]:o[i + 1]]) for i in range(self._M)]

    def fuse(self):
        self._push()
        M = self.stats()["M"]
        off = np.zeros(M + 1, _I32)
        self._chk(self.L.sz_download_fuse(self.h, capi.ptr(off, capi._ip), None))
        idx = np.zeros(max(off[-1], 1), _I32)
        self._chk(self.L.sz_download_fuse(self.h, capi.ptr(off, capi._ip), capi.ptr(idx, capi._ip)))
        return [list(idx[off[i]:off[i + 1]]) for i in range(M)]

    def pairs(self):
        n = self.stats()["n_pairs"]
        pi = np.zeros(max(n, 1), _I32); pj = np.zeros(max(n, 1), _I32)
        self._chk(self.L.sz_download_pairs(self.h, capi.ptr(pi, capi._ip), capi.ptr(pj, capi._ip)))
        return pi[:n], pj[:n]

    def boundary_vals(self):
        v = np.zeros(4)
        self._chk(self.L.sz_get_boundary_vals(self.h, capi.ptr(v)))
        return v

    def boundary_polys(self):
        """the four boundary polygons (N, S, E, W) as (5, 2) arrays, as _make_bounding_box_polygon orders them"""
        r = np.zeros(16)
        self._chk(self.L.sz_get_boundary_rects(self.h, capi.ptr(r)))
        out = []
        for k in range(4):
            x0, x1, y0, y1 = r[4 * k:4 * k + 4]
            out.append(np.array([[x0, y0], [x0, y1], [x1, y1], [x1, y0], [x0, y0]]))
        return out

    def which_vertices_match_points(self, points, region):
        """which_vertices_match_points(points, region) (floe_utils.jl:331-352) by the narrow phase's device routine: sorted
        1-based vertex indices"""
        p = np.ascontiguousarray(points, np.float64); r = np.ascontiguousarray(region, np.float64)
        px, py = np.ascontiguousarray(p[:, 0]), np.ascontiguousarray(p[:, 1])
        rx, ry = np.ascontiguousarray(r[:, 0]), np.ascontiguousarray(r[:, 1])
        idx = np.zeros(max(len(px), 1), _I32); n = C.c_int32(0)
        self._chk(self.L.sz_debug_match_vertices(self.h, len(px), capi.ptr(px), capi.ptr(py), len(rx), capi.ptr(rx), capi.ptr(ry),
                                                 capi.ptr(idx, capi._ip), C.byref(n)))
        return [int(v) + 1 for v in idx[:n.value]]

    def sample_fields(self, x, y):
        """the forcing kernels' in-bounds test and lattice sample at points (x, y): rows [in_bounds, uocn, vocn, hflx, uatm, vatm, line west, east,
        south, north (1-based), tx, ty]"""
        x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
        out = np.zeros((len(x), 12))
        self._chk(self.L.sz_debug_sample_fields(self.h, len(x), capi.ptr(x), capi.ptr(y), capi.ptr(out)))
        return out

    def pipelined(self):
        """the last run() batch ran as pipelined steps (two launches per timestep)"""
        return bool(self.L.sz_debug_pipelined(self.h))

    def warn_counts(self):
        s = self.stats()
        return np.array([s["warn_height"], s["warn_force"], s["warn_vel"], s["warn_xi"]], _I64)

    # ------------------------------------------------------------------ the reference's process API
    # ------------------------------------------------------------------ output path
    EUL_OUTPUTS = [
        "u_grid", "v_grid", "dudt_grid", "dvdt_grid", "overarea_grid", "mass_grid", "area_grid", "height_grid",
        "si_frac_grid", "stress_xx_grid", "stress_yx_grid", "stress_xy_grid", "stress_yy_grid", "stress_eig_grid",
        "strain_ux_grid", "strain_vx_grid", "strain_uy_grid", "strain_vy_grid",
    ]

    def eulerian_data(self, xg, yg, outputs=None):
        """calc_eulerian_data! (output.jl:793-914) over the rows the context holds: array [len(outputs), nx, ny]
        (writer.data[ix + 1, iy + 1, k + 1]); outputs: names out of EUL_OUTPUTS (get_known_grid_outputs), default all."""
        self._push()
        names = list(self.EUL_OUTPUTS if outputs is None else outputs)
        for n in names:
            if n not in self.EUL_OUTPUTS:
                raise SzError(f"{n} is not a known grid output")
        codes = np.array([self.EUL_OUTPUTS.index(n) for n in names], _I32)
        xg = np.ascontiguousarray(xg, np.float64); yg = np.ascontiguousarray(yg, np.float64)
        nx, ny = len(xg) - 1, len(yg) - 1
        out = np.zeros((max(len(names), 1), nx, ny))
        self._chk(self.L.sz_eulerian_data(self.h, nx, ny, capi.ptr(xg), capi.ptr(yg), len(names), capi.ptr(codes, capi._ip),
                                          capi.ptr(out)))
        return out[:len(names)]

    def write_grid_data(self, xg, yg, outputs=None):
        """write_grid_data! as timestep_sim! reaches it (simulation.jl:102-105): ghosts on, averages, ghosts off"""
        self.add_ghosts()
        try:
            return self.eulerian_data(xg, yg, outputs)
        finally:
            self.remove_ghosts()

    def simplify_check(self, max_vertices=30, min_floe_area=1e6, min_floe_height=0.1):
        """counts of what simplify_floes! (simplification.jl:339-378) would act on: remove, fuse, rings over
        max_vertices, floes under the minimum area / height; all zero = the pass is a no-op"""
        self._push()
        out = np.zeros(4, _I64)
        self._chk(self.L.sz_simplify_check(self.h, int(max_vertices), float(min_floe_area), float(min_floe_height),
                                           capi.ptr(out, capi._lp)))
        return dict(zip(("remove", "fuse", "over_max_vertices", "dissolve"), (int(v) for v in out)))

    def add_ghosts(self):
        self._push()
        self._chk(self.L.sz_add_ghosts(self.h)); self._host_stale = True

    def remove_ghosts(self, n_init=None):
        self._push()
        self._chk(self.L.sz_remove_ghosts(self.h)); self._host_stale = True

    def timestep_collisions(self, n_init, dt):
        self._push()
        self._chk(self.L.sz_timestep_collisions(self.h, int(n_init), int(dt))); self._host_stale = True

    def floe_floe_interaction(self, i, j, dt, max_overlap):
        self.collide_pairs([i], [j], dt, max_overlap)

    def collide_pairs(self, pi, pj, dt, max_overlap):
        self._push()
        pi = np.ascontiguousarray(pi, _I32); pj = np.ascontiguousarray(pj, _I32)
        self._chk(self.L.sz_collide_pairs(self.h, len(pi), capi.ptr(pi, capi._ip), capi.ptr(pj, capi._ip), int(dt),
                                          float(max_overlap)))
        self._host_stale = True

    def floe_domain_interaction(self, i, dt, max_overlap):
        self._push()
        self._chk(self.L.sz_collide_domain(self.h, int(dt), float(max_overlap))); self._host_stale = True

    def set_interactions(self, i, rows):
        """floe.interactions of floe i = rows (k x 7), set by hand as the reference's calc_stress! test does"""
        self._inter[int(i)] = np.ascontiguousarray(rows, np.float64).reshape(-1, 7)

    def _push_interactions(self):
        self._push()
        if not self._inter:
            return
        M = self._M
        off = np.zeros(M + 1, _I32)
        for i in range(M):
            off[i + 1] = off[i] + len(self._inter.get(i, ()))
        rows = np.concatenate([self._inter[i] for i in range(M) if i in self._inter]) if off[M] else np.zeros((1, 7))
        self._chk(self.L.sz_upload_interactions(self.h, capi.ptr(off, capi._ip), capi.ptr(rows)))
        self._inter = {}

    def calc_stress(self):
        self._push_interactions()
        self._chk(self.L.sz_calc_stress(self.h)); self._host_stale = True

    def calc_strain(self):
        self._push_interactions()
        self._chk(self.L.sz_calc_strain(self.h)); self._host_stale = True

    def calc_torque(self, i):
        """calc_torque! is fused into the interaction-list kernel; nothing to do."""

    def timestep_coupling(self):
        self._push()
        self._chk(self.L.sz_timestep_coupling(self.h)); self._host_stale = True

    def timestep_floe_properties(self, dt):
        self._push()
        self._chk(self.L.sz_timestep_floe_properties(self.h, int(dt))); self._host_stale = True

    def run(self, nsteps, tstep0, dt, coupling_dt=10, collisions_on=True, coupling_on=True, stop_on_tags=True):
        """nsteps x timestep_sim! with the state resident in HBM.  Returns the number of steps run: the batch ends
        after the first step that tags a floe remove / fuse (the reference runs simplify_floes! after every step,
        simulation.jl:205-214); stop_on_tags=False runs on regardless (measurement / soak runs)."""
        self._push()
        flags = (capi.COLLISIONS_ON if collisions_on else 0) | (capi.COUPLING_ON if coupling_on else 0)
        if not stop_on_tags:
            flags |= capi.NO_STOP
        done = C.c_int32(0)
        self._chk(self.L.sz_step(self.h, int(nsteps), int(tstep0), int(dt), int(coupling_dt), flags, C.byref(done)))
        self._host_stale = True
        return int(done.value)

    def timestep_sim(self, tstep, dt, coupling_dt=10, collisions_on=True, coupling_on=True):
        self.run(1, tstep, dt, coupling_dt, collisions_on, coupling_on)

    # ------------------------------------------------------------------ measurement
    def profile(self, on=True, only=None):
        """event-time the kernel classes (all, or the one named `only`, e.g. "narrow") from now on."""
        mode = (2 << capi.KERNEL_CLASS_NAMES.index(only)) if (on and only) else int(bool(on))
        self._chk(self.L.sz_profile_enable(self.h, mode)); self._chk(self.L.sz_profile_reset(self.h))

    def forcing_launch(self):
        """0 / 1 / 2: the forcings of the last resident batch ran in their own / the neighbour / the narrow launch (-1: none yet)"""
        w = np.zeros(1, np.int32)
        self._chk(self.L.sz_forcing_launch(self.h, w.ctypes.data_as(C.POINTER(C.c_int32))))
        return int(w[0])

    def narrow_kernel_name(self):
        """the dominant kernel's instantiation as a kernel trace names it, for the last batch"""
        buf = C.create_string_buffer(128)
        self._chk(self.L.sz_narrow_kernel_name(self.h, buf, 128))
        return buf.value.decode()

    def kernel_times(self):
        out = {}
        for k, name in enumerate(capi.KERNEL_CLASS_NAMES):
            ms = C.c_double(0); n = C.c_int64(0)
            self._chk(self.L.sz_kernel_time_ms(self.h, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out
