"""Synthetic floe fields for the benchmark configurations (BASELINE.json configs 2-5) and the
host-side sub-floe point generator.  Everything here is one-off setup on the host; the
per-timestep path never calls it.

Polygons: seeded (PCG64) star-shaped random polygons with 8-16 vertices and radii
r*(0.6 + 0.4*U), on a jittered lattice at a target concentration, clockwise rings (the
reference's tests and floe generator produce clockwise rings).  Sub-floe points follow the
reference's SubGridPointsGenerator (src/physical_processes/coupling.jl:169-321): points along
every edge plus the cell centres of a sub-grid that fall inside the floe.
"""
import numpy as np

from . import floe as floe_mod

KIND = {"open": 0, "periodic": 1, "collision": 2, "moving": 3}


def _points_in_ring(px, py, ring):
    x1, y1 = ring[:-1, 0][None, :], ring[:-1, 1][None, :]
    x2, y2 = ring[1:, 0][None, :], ring[1:, 1][None, :]
    X, Y = px[:, None], py[:, None]
    straddle = (y1 > Y) != (y2 > Y)
    with np.errstate(divide="ignore", invalid="ignore"):
        xi = x1 + (Y - y1) * (x2 - x1) / (y2 - y1)
    return (np.sum(straddle & (X < xi), axis=1) % 2) == 1


def _dist_outside_ring(px, py, ring):
    """distance of the points to the polygon, 0 for points inside it"""
    x1, y1 = ring[:-1, 0][None, :], ring[:-1, 1][None, :]
    dx, dy = ring[1:, 0][None, :] - x1, ring[1:, 1][None, :] - y1
    X, Y = px[:, None], py[:, None]
    t = np.clip(((X - x1) * dx + (Y - y1) * dy) / (dx * dx + dy * dy), 0.0, 1.0)
    d = np.sqrt(np.min((X - (x1 + t * dx)) ** 2 + (Y - (y1 + t * dy)) ** 2, axis=1))
    return np.where(_points_in_ring(px, py, ring), 0.0, d)


def subgrid_points(ring, cx, cy, dg):
    """generate_subfloe_points(::SubGridPointsGenerator, ...), coupling.jl:232-321 (centred ring)."""
    r = ring - np.array([cx, cy])
    xs, ys = [], []
    for k in range(len(r) - 1):
        x1, y1 = r[k]; x2, y2 = r[k + 1]
        dx, dy = x2 - x1, y2 - y1
        l = np.hypot(dx, dy)
        xs.append(x1); ys.append(y1)
        if l <= 2 * dg:
            if l > dg:
                xs.append(x1 + dx / 2); ys.append(y1 + dy / 2)
        else:
            if dx == 0:
                y1 += dg / 2 * np.sign(dy); y2 -= dg / 2 * np.sign(dy)
            elif dy == 0:
                x1 += dg / 2 * np.sign(dx); x2 -= dg / 2 * np.sign(dx)
            else:
                m = dy / dx
                xsft = np.sqrt(dg ** 2 / 4 / (1 + m ** 2)); ysft = m * xsft
                x1 += xsft; x2 -= xsft; y1 += ysft; y2 -= ysft
            l = np.hypot(x2 - x1, y2 - y1)
            ne = int(np.ceil(l / dg)) + 1
            xs.extend(np.linspace(x1, x2, ne)); ys.extend(np.linspace(y1, y2, ne))
    xmin, xmax, ymin, ymax = r[:, 0].min(), r[:, 0].max(), r[:, 1].min(), r[:, 1].max()
    nx = int(np.ceil((xmax - xmin) / dg)); ny = int(np.ceil((ymax - ymin) / dg))
    xi = np.zeros(1) if nx < 3 else np.linspace(xmin + dg / 2, xmax - dg / 2, nx)
    yi = np.zeros(1) if ny < 3 else np.linspace(ymin + dg / 2, ymax - dg / 2, ny)
    gx = np.tile(xi, len(yi)); gy = np.repeat(yi, len(xi))
    inside = _points_in_ring(gx, gy, r)
    return np.concatenate([np.array(xs), gx[inside]]), np.concatenate([np.array(ys), gy[inside]])


def voronoi_cells(rng, n_cells, L):
    """Voronoi cells of n_cells uniform random seed points, bounded by the box [0, L]^2 (the seeds are mirrored in the four
    walls, so every cell of an original seed ends at the box) -- the tessellation the reference's initialize_floe_field draws
    its floes from (floe_utils / floe.jl: generate_voronoi_coords; VoronoiCells.jl there, scipy's Qhull here).  Returns a list
    of closed clockwise rings."""
    from scipy.spatial import Voronoi
    p = rng.uniform(0.0, L, (n_cells, 2))
    pts = np.concatenate([p, p * [-1, 1], p * [-1, 1] + [2 * L, 0], p * [1, -1], p * [1, -1] + [0, 2 * L]])
    vor = Voronoi(pts)
    rings = []
    for i in range(n_cells):
        reg = vor.regions[vor.point_region[i]]
        if -1 in reg or len(reg) < 3:
            continue
        v = vor.vertices[reg]
        c = v.mean(0)
        order = np.argsort(-np.arctan2(v[:, 1] - c[1], v[:, 0] - c[0]))          # descending angle: clockwise
        v = np.clip(v[order], 0.0, L)                                             # (round-off of the mirror construction)
        keep = np.ones(len(v), bool)
        for k in range(len(v)):                                                   # Qhull may return coincident vertices
            if np.hypot(*(v[k] - v[k - 1])) < 1e-9 * L:
                keep[k] = False
        v = v[keep]
        if len(v) >= 3:
            rings.append(np.concatenate([v, v[:1]]))
    return rings


def make_config(n_floes=10000, seed=12345, concentration=0.8, spacing=2.0e4, walls=False, topography=False,
                ocean="uniform", dt=20, hmean=0.25, subgrid_per_floe=10.0, shape="star"):
    """Returns a plain dict describing one synthetic scenario (polygons, state, domain, fields).
    shape "star": random star-shaped polygons of 8-16 vertices on a jittered lattice (BASELINE.json configs[1]: "random-polygon
    floes (8-16 verts)"); "voronoi": n_floes cells drawn from a bounded Voronoi tessellation of n_floes / concentration seeds --
    the reference's own field generator (configs[0]): convex cells that TOUCH along whole edges, i.e. every contact starts degenerate."""
    rng = np.random.Generator(np.random.PCG64(seed))
    # mean star area = pi r^2 E[(0.6+0.4U)^2] ~ 0.6533 pi r^2  -> r from the target concentration
    r0 = spacing * np.sqrt(concentration / (0.6533 * np.pi))
    jit = 0.1 * spacing
    topo = []
    if not topography:
        n_side = int(np.ceil(np.sqrt(n_floes)))
        L = n_side * spacing
        cells = rng.permutation(n_side * n_side)[:n_floes]
        cells.sort()
        gx, gy = (cells % n_side).astype(float), (cells // n_side).astype(float)
        ccx = (gx + 0.5) * spacing + rng.uniform(-jit, jit, n_floes)
        ccy = (gy + 0.5) * spacing + rng.uniform(-jit, jit, n_floes)
    else:
        # the strait of examples/simple_strait.jl:24-28, scaled to the box: a coast-like wedge on the west and on the east side
        # and an island in the channel between them; floes only in the water -- the ones next to a coast reach a little into it,
        # so the floe-topography clip path has work from the first step on.  The lattice is sized so that n_floes cells are wet.
        strait = lambda s: [np.array([[6e4, 4e4], [6e4, 4.5e4], [6.5e4, 4.5e4], [6.5e4, 4e4], [6e4, 4e4]]) * s,
                            np.array([[0, 0.0], [0, 1e5], [2e4, 1e5], [3e4, 5e4], [2e4, 0], [0.0, 0.0]]) * s,
                            np.array([[8e4, 0], [7e4, 5e4], [8e4, 1e5], [1e5, 1e5], [1e5, 0], [8e4, 0]]) * s]
        n_side = int(np.ceil(np.sqrt(n_floes / 0.45)))
        while True:
            L = n_side * spacing
            topo = strait(L / 1e5)
            ax = (np.arange(n_side * n_side) % n_side + 0.5) * spacing; ay = (np.arange(n_side * n_side) // n_side + 0.5) * spacing
            wet = np.ones(n_side * n_side, bool)
            for t in topo:
                wet &= _dist_outside_ring(ax, ay, t) > 0.3 * r0 + jit
            if wet.sum() >= n_floes:
                break
            n_side += 1
        cells = np.nonzero(wet)[0]
        cells = cells[rng.permutation(len(cells))[:n_floes]]
        cells.sort()
        gx, gy = (cells % n_side).astype(float), (cells // n_side).astype(float)
        ccx = (gx + 0.5) * spacing + rng.uniform(-jit, jit, n_floes)
        ccy = (gy + 0.5) * spacing + rng.uniform(-jit, jit, n_floes)
    if shape == "voronoi":
        assert not topography, "the Voronoi generator fills the open box"
        rings = voronoi_cells(rng, int(np.ceil(n_floes / concentration)), L)
        assert len(rings) >= n_floes
        pick = np.sort(rng.permutation(len(rings))[:n_floes])
        rings = [rings[k] for k in pick]
        nv = np.array([len(r) - 1 for r in rings])
    else:
        nv = rng.integers(8, 17, n_floes)
    off = np.zeros(n_floes + 1, np.int32); off[1:] = np.cumsum(nv + 1)
    vx = np.zeros(off[-1]); vy = np.zeros(off[-1])
    if shape == "voronoi":
        vx[:] = np.concatenate([r[:, 0] for r in rings]); vy[:] = np.concatenate([r[:, 1] for r in rings])
    for i in range(n_floes if shape != "voronoi" else 0):
        n = nv[i]
        # jittered equally spaced angles: consecutive angles differ by < pi, so the ring is
        # star-shaped about its centre (hence simple); descending = clockwise
        th = (2 * np.pi / n) * (np.arange(n) + rng.uniform(-0.35, 0.35, n) + rng.uniform(0, 1))
        th = th[::-1]
        rad = r0 * (0.6 + 0.4 * rng.uniform(0, 1, n))
        x = ccx[i] + rad * np.cos(th); y = ccy[i] + rad * np.sin(th)
        o = off[i]
        vx[o:o + n] = x; vy[o:o + n] = y; vx[o + n] = x[0]; vy[o + n] = y[0]
    u = rng.uniform(-0.1, 0.1, n_floes); v = rng.uniform(-0.1, 0.1, n_floes); xi = rng.uniform(-1e-6, 1e-6, n_floes)
    kinds = ["collision"] * 4 if walls else ["periodic"] * 4
    dgrid = spacing / 4.0
    Nx = Ny = int(round(L / dgrid))
    xl = np.linspace(0.0, L, Nx + 1)
    if ocean == "uniform":                   # examples/uniform_flow.jl:14-15
        uo = np.full((Nx + 1, Ny + 1), 0.1); vo = np.zeros((Nx + 1, Ny + 1))
    elif ocean == "shear":                   # examples/shear_flow.jl:16-25: u 0 -> 0.5 -> 0 across y, v = 0
        prof = 0.5 * (1.0 - np.abs(2.0 * xl / L - 1.0))
        uo = np.repeat(prof[None, :], Nx + 1, 0); vo = np.zeros((Nx + 1, Ny + 1))
    elif ocean == "strait":                  # examples/simple_strait.jl:14
        uo = np.zeros((Nx + 1, Ny + 1)); vo = np.full((Nx + 1, Ny + 1), -0.3)
    elif ocean == "converge_diverge":        # examples/converge_diverge_flow.jl:16-23: 0.1 -> 0.6 -> 0.1 across x
        prof = 0.1 + 0.5 * (1.0 - np.abs(2.0 * xl / L - 1.0))
        uo = np.repeat(prof[:, None], Ny + 1, 1); vo = np.zeros((Nx + 1, Ny + 1))
    else:
        raise ValueError(ocean)
    cfg = dict(n_floes=n_floes, seed=seed, L=L, kinds=kinds, vert_off=off, vx=vx, vy=vy,
               height=np.full(n_floes, hmean), u=u, v=v, xi=xi, dt=dt, Nx=Nx, Ny=Ny,
               uo=uo, vo=vo, hf=np.zeros((Nx + 1, Ny + 1)), ua=np.zeros((Nx + 1, Ny + 1)), va=np.zeros((Nx + 1, Ny + 1)),
               topography=topo, dg=2.0 * r0 / subgrid_per_floe)
    d = floe_mod.derive(off, vx, vy, cfg["height"])
    cfg["derived"] = d
    # examples/uniform_flow.jl:37: E = 1.5e3*(mean(sqrt(area)) + min(sqrt(area)))
    sq = np.sqrt(d["area"])
    cfg["E"] = 1.5e3 * (sq.mean() + sq.min())
    so = np.zeros(n_floes + 1, np.int32); sxs = []; sys_ = []
    for i in range(n_floes):
        ring = np.stack([vx[off[i]:off[i + 1]], vy[off[i]:off[i + 1]]], 1)
        sx, sy = subgrid_points(ring, d["cx"][i], d["cy"][i], cfg["dg"])
        so[i + 1] = so[i] + len(sx); sxs.append(sx); sys_.append(sy)
    cfg["sub_off"] = so; cfg["sx"] = np.concatenate(sxs); cfg["sy"] = np.concatenate(sys_)
    return cfg


def build_world(w, cfg):
    """Fills a World-like object (subzero_jl_amd.World or the oracle binding) from a config."""
    w.set_consts(E=cfg["E"])
    w.set_settings()
    L = cfg["L"]
    w.set_domain([KIND[k] for k in cfg["kinds"]], 0.0, L, 0.0, L)
    if cfg["topography"]:
        w.set_topography(cfg["topography"])
    w.set_grid_fields(cfg["Nx"], cfg["Ny"], 0.0, L, 0.0, L, cfg["uo"], cfg["vo"], cfg["hf"], cfg["ua"], cfg["va"])
    off, vx, vy = cfg["vert_off"], cfg["vx"], cfg["vy"]
    n = cfg["n_floes"]
    if hasattr(w, "load_columns"):
        d = cfg["derived"]
        cols = dict(cx=d["cx"], cy=d["cy"], rmax=d["rmax"], area=d["area"], height=d["height"], mass=d["mass"],
                    moment=d["moment"], u=cfg["u"], v=cfg["v"], xi=cfg["xi"], vert_off=off, vx=vx, vy=vy)
        w.load_columns(cols)
        w.set_subpoints_csr(cfg["sub_off"], cfg["sx"], cfg["sy"])
    else:
        so = cfg["sub_off"]
        for i in range(n):
            w.add_floe(np.stack([vx[off[i]:off[i + 1]], vy[off[i]:off[i + 1]]], 1), cfg["height"][i])
            w.set_subpoints(i, cfg["sx"][so[i]:so[i + 1]], cfg["sy"][so[i]:so[i + 1]])
        w.set("u", cfg["u"]); w.set("v", cfg["v"]); w.set("xi", cfg["xi"])
    return w
