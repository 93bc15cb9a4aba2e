"""Host-side floe setup (runs once per floe, outside the per-timestep path): what the reference's
Floe constructor derives from a coordinate ring (src/simulation_components/floe.jl:144-200).

The sums run in ring order, one IEEE operation at a time, so the derived columns are the
values the reference's formulas give for the same ring:
  centroid  GO.centroid (area-weighted, raw coordinates)
  area      GO.area (unsigned shoelace)
  mass      area * height * rho_i
  moment    _calc_moment_inertia (src/floe_utils.jl:273-298; the centroid enters `wi` twice, literal)
  rmax      calc_max_radius (src/floe_utils.jl:301-313)
Vectorised over floes: the loop runs over the ring position, every floe advances in lockstep.
"""
import numpy as np


def valid_ring(coords):
    """valid_ringvec!, src/floe_utils.jl:10-17: drop adjacent duplicates, close the ring."""
    c = np.asarray(coords, dtype=np.float64)
    keep = np.ones(len(c), bool)
    keep[:-1] = np.any(c[:-1] != c[1:], axis=1)
    c = c[keep]
    if np.any(c[0] != c[-1]):
        c = np.vstack([c, c[:1]])
    assert len(c) > 3, "Polygon needs at least 3 distinct points."
    return c


def _padded(vert_off, vx, vy):
    vert_off = np.asarray(vert_off)
    n = np.diff(vert_off)
    nmax = int(n.max())
    M = len(n)
    X = np.zeros((M, nmax)); Y = np.zeros((M, nmax))
    idx = vert_off[:-1, None] + np.arange(nmax)[None, :]
    mask = np.arange(nmax)[None, :] < n[:, None]
    X[mask] = vx[idx[mask]]; Y[mask] = vy[idx[mask]]
    return X, Y, n, mask


def derive(vert_off, vx, vy, height, rho_i=920.0):
    """Returns dict(cx, cy, area, mass, moment, rmax) for closed rings in CSR form."""
    vx = np.asarray(vx, np.float64); vy = np.asarray(vy, np.float64)
    X, Y, n, mask = _padded(vert_off, vx, vy)
    M, nmax = X.shape
    height = np.broadcast_to(np.asarray(height, np.float64), (M,)).copy()
    # ---- GO.centroid_and_area / GO._signed_area
    xc = np.zeros(M); yc = np.zeros(M); a2 = np.zeros(M); sa = np.zeros(M)
    for k in range(1, nmax):
        live = mask[:, k]
        p1x, p1y, p2x, p2y = X[:, k - 1], Y[:, k - 1], X[:, k], Y[:, k]
        ac = p1x * p2y - p2x * p1y
        sarea = p1x * p2y - p1y * p2x
        a2 = np.where(live, a2 + ac, a2)
        xc = np.where(live, xc + (p1x + p2x) * ac, xc)
        yc = np.where(live, yc + (p1y + p2y) * ac, yc)
        sa = np.where(live, sa + sarea, sa)
    last = n - 1
    lx, ly = X[np.arange(M), last], Y[np.arange(M), last]
    sa = sa + (lx * Y[:, 0] - ly * X[:, 0])          # closing edge of _signed_area (zero for closed rings)
    area = np.abs(sa / 2.0)
    a_half = a2 / 2.0
    cx = xc / (6.0 * a_half); cy = yc / (6.0 * a_half)
    # ---- calc_max_radius
    rs = np.zeros(M)
    for k in range(nmax):
        x = X[:, k] - cx; y = Y[:, k] - cy
        r = x * x + y * y
        rs = np.where(mask[:, k] & (r > rs), r, rs)
    rmax = np.sqrt(rs)
    # ---- _calc_moment_inertia
    Ixx = np.zeros(M); Iyy = np.zeros(M)
    x1 = X[:, 0] - cx; y1 = Y[:, 0] - cy
    for k in range(1, nmax):
        live = mask[:, k]
        x2 = X[:, k] - cx; y2 = Y[:, k] - cy
        wi = (x1 - cx) * (y2 - cy) - (x2 - cx) * (y1 - cy)
        Ixx = np.where(live, Ixx + wi * (y1 * y1 + y1 * y2 + y2 * y2), Ixx)
        Iyy = np.where(live, Iyy + wi * (x1 * x1 + x1 * x2 + x2 * x2), Iyy)
        x1 = np.where(live, x2, x1); y1 = np.where(live, y2, y1)
    Ixx = Ixx * (1.0 / 12.0); Iyy = Iyy * (1.0 / 12.0)
    moment = np.abs(Ixx + Iyy) * height * rho_i
    mass = area * height * rho_i
    return dict(cx=cx, cy=cy, area=area, mass=mass, moment=moment, rmax=rmax, height=height)


def topography_props(ring):
    """TopographyElement (domain_components/topography.jl:66-72): centroid and rmax."""
    r = np.asarray(ring, np.float64)
    d = derive(np.array([0, len(r)]), r[:, 0].copy(), r[:, 1].copy(), 1.0)
    return float(d["cx"][0]), float(d["cy"][0]), float(d["rmax"][0])


def boundary_rects(x0, xf, y0, yf):
    """_boundary_info_from_extent (boundaries.jl:29,65,102,139): rectangles {xmin,xmax,ymin,ymax}
    and wall values in the order N, S, E, W."""
    dx, dy = (xf - x0) / 2, (yf - y0) / 2
    rects = np.array([
        [x0 - dx, xf + dx, yf, yf + dy],
        [x0 - dx, xf + dx, y0 - dy, y0],
        [xf, xf + dx, y0 - dy, yf + dy],
        [x0 - dx, x0, y0 - dy, yf + dy],
    ], dtype=np.float64)
    vals = np.array([yf, y0, xf, x0], dtype=np.float64)
    return rects, vals
