"""ctypes binding of the CPU ORACLE (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py.  The product package (subzero.jl_amd/) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FIELDS = [
    "cx", "cy", "rmax", "area", "height", "mass", "moment", "alpha", "u", "v", "xi",
    "p_dxdt", "p_dydt", "p_dalphadt", "p_dudt", "p_dvdt", "p_dxidt",
    "fxOA", "fyOA", "trqOA", "hflx_factor", "overarea",
    "coll_fx", "coll_fy", "coll_trq",
    "sa11", "sa12", "sa21", "sa22", "si11", "si12", "si21", "si22",
    "e11", "e12", "e21", "e22",
]
FIELD_ID = {n: i for i, n in enumerate(FIELDS)}

OPEN, PERIODIC, COLLISION, MOVING = 0, 1, 2, 3
NORTH, SOUTH, EAST, WEST = 0, 1, 2, 3
ACTIVE, REMOVE, FUSE = 1, 2, 3

# grid outputs of calc_eulerian_data! in the order of ORC_EUL_* (output.jl:855-905)
EUL_OUTPUTS = [
    "u_grid", "v_grid", "dudt_grid", "dvdt_grid", "overarea_grid", "mass_grid", "area_grid", "height_grid",
    "si_frac_grid", "stress_xx_grid", "stress_yx_grid", "stress_xy_grid", "stress_yy_grid", "stress_eig_grid",
    "strain_ux_grid", "strain_vx_grid", "strain_uy_grid", "strain_vy_grid",
]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


def build(force=False):
    """Compile oracle/liborc.so with the committed Makefile (gcc)."""
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_geom.c", "orc_world.c", "orc.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_create.restype = C.c_void_p
        L.orc_add_floe.restype = C.c_int
        for name in ("orc_num_floes", "orc_total_ring_points", "orc_total_interactions",
                     "orc_total_ghost_links", "orc_total_fuse", "orc_num_pairs",
                     "orc_clip_flat", "orc_ipoints_flat", "orc_cell_count", "orc_shift_cell_idx", "orc_which_vertices_match_points",
                     "orc_in_bounds", "orc_find_interp_knots"):
            getattr(L, name).restype = C.c_int
    return _LIB


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t=_dp):
    return a.ctypes.data_as(t)


def clip(a, b, max_regions=16, max_pts=4096):
    """intersect_polys(a, b): list of closed rings ((n,2) arrays). a, b: (n,2) closed rings."""
    a = _d(a); b = _d(b)
    ax, ay = _d(a[:, 0]), _d(a[:, 1]); bx, by = _d(b[:, 0]), _d(b[:, 1])
    off = np.zeros(max_regions + 1, np.int32); rx = np.zeros(max_pts); ry = np.zeros(max_pts)
    n = lib().orc_clip_flat(len(ax), _p(ax), _p(ay), len(bx), _p(bx), _p(by), max_regions, max_pts,
                            _p(off, _ip), _p(rx), _p(ry))
    assert n >= 0, "oracle clip capacity exceeded"
    return [np.stack([rx[off[k]:off[k + 1]], ry[off[k]:off[k + 1]]], 1) for k in range(n)]


def which_vertices_match_points(points, region):
    """which_vertices_match_points(points, region) (floe_utils.jl:331-352): sorted 1-based vertex indices"""
    p = _d(points); r = _d(region)
    px, py, rx, ry = _d(p[:, 0]), _d(p[:, 1]), _d(r[:, 0]), _d(r[:, 1])
    idx = np.zeros(max(len(px), 1), np.int32)
    n = lib().orc_which_vertices_match_points(len(px), _p(px), _p(py), len(rx), _p(rx), _p(ry), _p(idx, _ip))
    return [int(v) + 1 for v in idx[:n]]


def find_interp_knots(point_idx, ncells, g0, dg, L, dd, periodic):
    """find_interp_knots (coupling.jl:702-797): (knots, 1-based knot_idx)"""
    pts = np.ascontiguousarray(point_idx, dtype=np.int32)
    cap = 4 * (ncells + 2 * dd + 8)
    knots = np.zeros(cap); idx = np.zeros(cap, np.int32)
    n = lib().orc_find_interp_knots(len(pts), _p(pts, _ip), int(ncells), C.c_double(g0), C.c_double(dg), C.c_double(L), int(dd), int(periodic),
                                    cap, _p(knots), _p(idx, _ip))
    assert 0 <= n <= cap
    return knots[:n].tolist(), idx[:n].tolist()


def intersection_points(a, b, max_pts=1024):
    a = _d(a); b = _d(b)
    ax, ay = _d(a[:, 0]), _d(a[:, 1]); bx, by = _d(b[:, 0]), _d(b[:, 1])
    px = np.zeros(max_pts); py = np.zeros(max_pts)
    n = lib().orc_ipoints_flat(len(ax), _p(ax), _p(ay), len(bx), _p(bx), _p(by), max_pts, _p(px), _p(py))
    return np.stack([px[:n], py[:n]], 1)


class World:
    """Mirror of the reference's Simulation/Model state restricted to the hot path."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_create())

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    # ---- setup
    def set_consts(self, E=6e6, nu=0.3, mu=0.2, rho_o=1027.0, rho_a=1.2, Cd_io=3e-3, Cd_ia=1e-3,
                   f=1.4e-4, turn_theta=15 * np.pi / 180):
        self.L.orc_set_consts(self.h, *(C.c_double(v) for v in (E, nu, mu, rho_o, rho_a, Cd_io, Cd_ia, f, turn_theta)))

    def set_settings(self, floe_floe_max_overlap=0.55, floe_domain_max_overlap=0.75, rho_i=920.0,
                     max_floe_height=10.0, maximum_xi=1e-5, lam=0.2, coupling_dd=1):
        self.L.orc_set_settings(self.h, *(C.c_double(v) for v in (floe_floe_max_overlap, floe_domain_max_overlap,
                                                                 rho_i, max_floe_height, maximum_xi, lam)),
                                C.c_int(coupling_dd))

    def set_domain(self, kinds, x0, xf, y0, yf, bu=None, bv=None):
        k = np.asarray(kinds, np.int32)
        bu = _d(bu if bu is not None else np.zeros(4)); bv = _d(bv if bv is not None else np.zeros(4))
        self.L.orc_set_domain_extent(self.h, _p(k, _ip), C.c_double(x0), C.c_double(xf), C.c_double(y0),
                                     C.c_double(yf), _p(bu), _p(bv))

    def set_topography(self, rings):
        off = np.zeros(len(rings) + 1, np.int32)
        for i, r in enumerate(rings):
            off[i + 1] = off[i] + len(r)
        xy = np.concatenate([_d(r) for r in rings], 0) if rings else np.zeros((0, 2))
        x, y = _d(xy[:, 0]), _d(xy[:, 1])
        self.L.orc_set_topography(self.h, len(rings), _p(off, _ip), _p(x), _p(y))

    def set_grid_fields(self, Nx, Ny, x0, xf, y0, yf, uo, vo, hflx, ua, va):
        arrs = [_d(np.broadcast_to(a, (Nx + 1, Ny + 1))) for a in (uo, vo, hflx, ua, va)]
        self.L.orc_set_grid_fields(self.h, Nx, Ny, C.c_double(x0), C.c_double(xf), C.c_double(y0),
                                   C.c_double(yf), *(_p(a) for a in arrs))

        self._grid = (Nx, Ny)

    # ---- two-way coupling (coupling.jl:1617-1680); off by default like CouplingSettings()
    def set_two_way(self, on=True, Cd_ao=1.25e-3, k=2.14, L=2.93e5, dt=10):
        self.L.orc_set_two_way(self.h, int(on), C.c_double(Cd_ao), C.c_double(k), C.c_double(L), int(dt))

    def set_temps(self, t_ocn, t_atm):
        Nx, Ny = self._grid
        a, b = (_d(np.broadcast_to(t, (Nx + 1, Ny + 1))) for t in (t_ocn, t_atm))
        self.L.orc_set_temps(self.h, _p(a), _p(b))

    def ocean_stress(self):
        """tau_x, tau_y, si_frac, hflx_factor on the (Nx+1) x (Ny+1) grid-line lattice"""
        Nx, Ny = self._grid
        out = [np.zeros((Nx + 1, Ny + 1)) for _ in range(4)]
        self.L.orc_get_ocean_stress(self.h, *(_p(a) for a in out))
        return out

    def floe_to_grid_info(self, floeidx, xidx, yidx, tx, ty):
        self.L.orc_floe_to_grid_info(self.h, int(floeidx), int(xidx), int(yidx), C.c_double(tx), C.c_double(ty))

    def cell_entries(self, xidx, yidx):
        """rows (floeidx, dx, dy, sum tx, sum ty, npoints) of centre cell (xidx, yidx), 1-based"""
        n = self.L.orc_cell_count(self.h, int(xidx), int(yidx))
        out = np.zeros((n, 6))
        for k in range(n):
            row = np.zeros(6); self.L.orc_cell_entry(self.h, int(xidx), int(yidx), k, _p(row)); out[k] = row
        return out

    def in_bounds(self, x, y, per_x, per_y):
        """in_bounds (coupling.jl:494-597) as the coupling evaluates it per sub-floe point; per_x: the east/west pair is periodic"""
        return bool(self.L.orc_in_bounds(self.h, C.c_double(x), C.c_double(y), int(per_x), int(per_y)))

    def sample_lines(self, x, y, per_x, per_y):
        """the 1-based grid lines (west, east, south, north) and the weights (tx, ty) the lattice sample blends at (x, y)"""
        lines = np.zeros(4, np.int32); t = np.zeros(2)
        self.L.orc_sample_lines(self.h, C.c_double(x), C.c_double(y), int(per_x), int(per_y), _p(lines, _ip), _p(t))
        return lines.tolist(), t.tolist()

    def sample_fields(self, x, y, per_x, per_y):
        out = np.zeros(5)
        self.L.orc_sample_fields(self.h, C.c_double(x), C.c_double(y), int(per_x), int(per_y), _p(out))
        return out

    def center_cell_coords(self, xidx, yidx, ns_periodic, ew_periodic):
        out = np.zeros(4)
        self.L.orc_center_cell_coords(self.h, int(xidx), int(yidx), int(ns_periodic), int(ew_periodic), _p(out))
        return out

    def add_floe(self, coords, height):
        c = _d(coords)
        x, y = _d(c[:, 0]), _d(c[:, 1])
        return self.L.orc_add_floe(self.h, len(x), _p(x), _p(y), C.c_double(height))

    def set_subpoints(self, i, sx, sy):
        sx, sy = _d(sx), _d(sy)
        self.L.orc_set_subpoints(self.h, int(i), len(sx), _p(sx), _p(sy))

    # ---- state access
    @property
    def M(self):
        return self.L.orc_num_floes(self.h)

    def get(self, name):
        out = np.zeros(self.M)
        self.L.orc_get_field(self.h, FIELD_ID[name], _p(out))
        return out

    def set(self, name, vals):
        v = _d(np.broadcast_to(vals, (self.M,)))
        self.L.orc_set_field(self.h, FIELD_ID[name], _p(v))

    def ids(self):
        M = self.M
        i = np.zeros(M, np.int64); g = np.zeros(M, np.int64); s = np.zeros(M, np.int32)
        self.L.orc_get_ids(self.h, _p(i, _lp), _p(g, _lp), _p(s, _ip))
        return i, g, s

    def set_ids(self, ids):
        v = np.ascontiguousarray(ids, np.int64)
        self.L.orc_set_ids(self.h, _p(v, _lp))

    def set_status(self, st):
        v = np.ascontiguousarray(st, np.int32)
        self.L.orc_set_status(self.h, _p(v, _ip))

    def rings(self):
        M = self.M; T = self.L.orc_total_ring_points(self.h)
        off = np.zeros(M + 1, np.int32); x = np.zeros(T); y = np.zeros(T)
        self.L.orc_get_rings(self.h, _p(off, _ip), _p(x), _p(y))
        return off, x, y

    def ring(self, i):
        off, x, y = self.rings()
        return np.stack([x[off[i]:off[i + 1]], y[off[i]:off[i + 1]]], 1)

    def interactions(self):
        M = self.M; T = self.L.orc_total_interactions(self.h)
        off = np.zeros(M + 1, np.int32); rows = np.zeros((max(T, 1), 7))
        self.L.orc_get_interactions(self.h, _p(off, _ip), _p(rows))
        return off, rows[:T]

    def inter(self, i):
        off, rows = self.interactions()
        return rows[off[i]:off[i + 1]]

    def set_interactions(self, i, rows):
        """floe.interactions = rows (k x 7: floeidx, xforce, yforce, xpoint, ypoint, torque, overlap)"""
        r = np.ascontiguousarray(rows, np.float64).reshape(-1, 7)
        self.L.orc_set_interactions(self.h, int(i), int(len(r)), _p(r))

    def calc_stress(self, i=None):
        for k in (range(self.M) if i is None else [i]):
            self.L.orc_calc_stress(self.h, int(k))

    def calc_strain(self, i=None):
        for k in (range(self.M) if i is None else [i]):
            self.L.orc_calc_strain(self.h, int(k))

    def ghosts(self):
        M = self.M; T = self.L.orc_total_ghost_links(self.h)
        off = np.zeros(M + 1, np.int32); idx = np.zeros(max(T, 1), np.int32)
        self.L.orc_get_ghosts(self.h, _p(off, _ip), _p(idx, _ip))
        return [list(idx[off[i]:off[i + 1]]) for i in range(M)]

    def fuse(self):
        M = self.M; T = self.L.orc_total_fuse(self.h)
        off = np.zeros(M + 1, np.int32); idx = np.zeros(max(T, 1), np.int32)
        self.L.orc_get_fuse(self.h, _p(off, _ip), _p(idx, _ip))
        return [list(idx[off[i]:off[i + 1]]) for i in range(M)]

    def pairs(self):
        n = self.L.orc_num_pairs(self.h)
        pi = np.zeros(max(n, 1), np.int32); pj = np.zeros(max(n, 1), np.int32)
        self.L.orc_get_pairs(self.h, _p(pi, _ip), _p(pj, _ip))
        return pi[:n], pj[:n]

    def boundary_vals(self):
        v = np.zeros(4)
        self.L.orc_get_boundary_vals(self.h, _p(v))
        return v

    def boundary_polys(self):
        """the four boundary polygons (N, S, E, W) as (5, 2) arrays"""
        v = np.zeros(40)
        self.L.orc_get_boundary_polys(self.h, _p(v))
        return [np.stack([v[k * 10:k * 10 + 5], v[k * 10 + 5:k * 10 + 10]], 1) for k in range(4)]

    def warn_counts(self):
        v = np.zeros(4, np.int64)
        self.L.orc_get_warn_counts(self.h, _p(v, _lp))
        return v

    def set_threads(self, n):
        self.L.orc_set_threads(self.h, int(n))

    PHASES = ["add_ghosts", "pair_loop_all_pairs_circles", "dict_pass_serial", "floe_floe_and_domain_interactions",
              "mirror_ghost_fold_totals_serial", "timestep_coupling_serial", "timestep_floe_properties"]

    def phase_times(self, reset=True):
        """wall seconds per phase of the steps run since the last reset (where the CPU path spends its time)"""
        out = np.zeros(8)
        self.L.orc_get_phase_times(self.h, _p(out))
        if reset:
            self.L.orc_reset_phase_times(self.h)
        return {n: float(out[k]) for k, n in enumerate(self.PHASES)}

    # ---- the reference's process API
    def eulerian_data(self, xg, yg):
        """calc_eulerian_data! (output.jl:793-914): array [len(EUL_OUTPUTS), nx, ny]."""
        xg, yg = _d(xg), _d(yg)
        nx, ny = len(xg) - 1, len(yg) - 1
        out = np.zeros((len(EUL_OUTPUTS), nx, ny))
        self.L.orc_calc_eulerian_data(self.h, C.c_int(nx), C.c_int(ny), _p(xg), _p(yg), _p(out))
        return out

    def simplify_check(self, max_vertices, min_floe_area, min_floe_height):
        out = np.zeros(4, dtype=np.int64)
        self.L.orc_simplify_check(self.h, C.c_int(max_vertices), C.c_double(min_floe_area), C.c_double(min_floe_height),
                                  _p(out, _lp))
        return out

    def add_ghosts(self):
        self.L.orc_add_ghosts(self.h)

    def remove_ghosts(self, n_init):
        self.L.orc_remove_ghosts(self.h, int(n_init))

    def timestep_collisions(self, n_init, dt):
        self.L.orc_timestep_collisions(self.h, int(n_init), int(dt))

    def floe_floe_interaction(self, i, j, dt, max_overlap):
        self.L.orc_floe_floe_interaction(self.h, int(i), int(j), int(dt), C.c_double(max_overlap))

    def floe_domain_interaction(self, i, dt, max_overlap):
        self.L.orc_floe_domain_interaction(self.h, int(i), int(dt), C.c_double(max_overlap))

    def calc_torque(self, i):
        self.L.orc_calc_torque(self.h, int(i))

    def timestep_coupling(self):
        self.L.orc_timestep_coupling(self.h)

    def timestep_floe_properties(self, dt):
        self.L.orc_timestep_floe_properties(self.h, int(dt))

    def timestep_sim(self, tstep, dt, coupling_dt=10, collisions_on=True, coupling_on=True):
        self.L.orc_timestep_sim(self.h, int(tstep), int(dt), int(coupling_dt), int(collisions_on), int(coupling_on))
