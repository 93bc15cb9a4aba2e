"""Parity helpers: compare the HIP engine's state with the CPU oracle's on the same inputs.

Tolerances.  BASELINE.json asks for "forces within 1e-10 relative of reference".  Interaction rows and the
per-floe collision totals are therefore checked PER ELEMENT,

    |a - b| <= rtol * |b| + atol,

never against the largest value of the column (a small contact next to a 1e9 N one must be right too).  The
absolute floor `atol` is the round-off floor of the reference's own arithmetic, stated from the field:
GO.area / GO.centroid are shoelace sums over ABSOLUTE coordinates (collisions.jl:360,178), so an overlap area
carries an absolute error of a few eps * Lc^2 (Lc = coordinate magnitude of the contact), whatever its size:

    area_floor  = 64 * eps * Lc^2                       (0.06 m^2 in a 2000 km box; overlaps are 1e4..1e7 m^2)
    force_floor = max_rows(|F| / overlap) * area_floor  (a force is force_factor * area, collisions.jl:69)
    torque_floor= force_floor * max(rmax)               (lever arm, collisions.jl:673-686)

Totals of a floe are sums of its rows with cancellation: their `|b|` is the sum of the floe's |row forces|.
State columns (positions, velocities, stresses ...) are sums and differences of terms of the column's scale and
are compared on that scale (max-norm): a velocity that passes through zero has no meaningful relative error.
Every helper returns / reports the worst element so that a failure names the floe and row.
"""
import numpy as np

EPS = np.finfo(float).eps

SCALARS = ["cx", "cy", "alpha", "u", "v", "xi", "height", "mass", "moment", "p_dxdt", "p_dydt", "p_dalphadt",
           "p_dudt", "p_dvdt", "p_dxidt", "fxOA", "fyOA", "trqOA", "hflx_factor", "overarea",
           "coll_fx", "coll_fy", "coll_trq",
           "sa11", "sa12", "sa21", "sa22", "si11", "si12", "si21", "si22", "e11", "e12", "e21", "e22"]
TOTALS = ("coll_fx", "coll_fy", "coll_trq")


def relerr(a, b):
    """max-norm relative error: for state columns (see the module docstring), not for forces"""
    a = np.asarray(a, float); b = np.asarray(b, float)
    if a.size == 0 and b.size == 0:
        return 0.0
    scale = max(np.max(np.abs(b)), 1e-300)
    return float(np.max(np.abs(a - b)) / scale)


def worst_element(a, b, rtol, atol, scale=None):
    """per element |a-b| <= rtol*scale + atol (scale defaults to |b|).  Returns (ratio, index): ratio = largest
    |a-b| / (rtol*scale + atol), > 1 means failure; index of that element."""
    a = np.asarray(a, float).ravel(); b = np.asarray(b, float).ravel()
    if a.size == 0:
        return 0.0, -1
    s = np.abs(b) if scale is None else np.asarray(scale, float).ravel()
    tol = rtol * s + atol
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(tol > 0, np.abs(a - b) / tol, np.where(a == b, 0.0, np.inf))
    ratio = np.where(np.isnan(ratio), np.inf, ratio)
    k = int(np.argmax(ratio))
    return float(ratio[k]), k


def assert_elementwise(name, a, b, rtol, atol, scale=None):
    r, k = worst_element(a, b, rtol, atol, scale)
    a = np.asarray(a, float).ravel(); b = np.asarray(b, float).ravel()
    assert r <= 1.0, (f"{name}: element {k}: got {a[k]!r}, oracle {b[k]!r}, |diff| {abs(a[k] - b[k]):.3e} is {r:.2f}x the "
                      f"tolerance (rtol {rtol:g}, atol {np.max(atol) if np.ndim(atol) else atol:.3e})")
    return r


def force_floors(orows, rmax_max):
    """the absolute floors of the module docstring from the oracle's rows (k x 7)"""
    if len(orows) == 0:
        return dict(area=0.0, force=0.0, torque=0.0, Lc=1.0)
    Lc = max(float(np.max(np.abs(orows[:, 3:5]))), 1.0)
    area_floor = 64 * EPS * Lc * Lc
    fmag = np.hypot(orows[:, 1], orows[:, 2])
    ok = orows[:, 6] > 0
    per_area = float(np.max(fmag[ok] / orows[ok, 6])) if np.any(ok) else 0.0
    f = per_area * area_floor
    return dict(area=area_floor, force=f, torque=f * float(rmax_max), Lc=Lc)


def compare_pairs(hw, ow):
    hi, hj = hw.pairs(); oi, oj = ow.pairs()
    assert len(hi) == len(oi), (len(hi), len(oi))
    assert np.array_equal(hi, oi) and np.array_equal(hj, oj), "overlap-pair indices differ"
    return len(hi)


def compare_interactions(hw, ow, rtol, report=None):
    """interaction rows of every floe, per element (module docstring).  Returns the worst ratio per column (<= 1)."""
    hoff, hrows = hw.interactions(); ooff, orows = ow.interactions()
    n = min(len(hoff), len(ooff))
    assert np.array_equal(hoff[:n], ooff[:n]), "interaction row counts differ"
    m = hoff[n - 1]
    hrows, orows = hrows[:m], orows[:m]
    assert np.array_equal(hrows[:, 0], orows[:, 0]), "partner indices differ"
    fl = force_floors(orows, np.max(ow.get("rmax")) if m else 1.0)
    errs = {}
    errs["xforce"] = assert_elementwise("xforce", hrows[:, 1], orows[:, 1], rtol, fl["force"])
    errs["yforce"] = assert_elementwise("yforce", hrows[:, 2], orows[:, 2], rtol, fl["force"])
    errs["torque"] = assert_elementwise("torque", hrows[:, 5], orows[:, 5], rtol, fl["torque"])
    errs["overlap"] = assert_elementwise("overlap", hrows[:, 6], orows[:, 6], rtol, fl["area"])
    # force points are coordinates: relative to the coordinate magnitude of the contact
    errs["point"] = max(assert_elementwise("xpoint", hrows[:, 3], orows[:, 3], rtol, rtol * fl["Lc"]),
                        assert_elementwise("ypoint", hrows[:, 4], orows[:, 4], rtol, rtol * fl["Lc"]))
    if report is not None:
        report.update(floors=fl, worst=errs)
    return errs


def compare_totals(hw, ow, rtol):
    """collision_force / collision_trq per floe: |a-b| <= rtol * sum_rows|F| + floor, per floe"""
    ooff, orows = ow.interactions()
    M = ow.M
    fl = force_floors(orows, np.max(ow.get("rmax")) if len(orows) else 1.0)
    owner = np.repeat(np.arange(len(ooff) - 1), np.diff(ooff))[:len(orows)]
    sF = np.zeros(M); sT = np.zeros(M); cnt = np.zeros(M)
    if len(orows):
        np.add.at(sF, owner, np.hypot(orows[:, 1], orows[:, 2])); np.add.at(sT, owner, np.abs(orows[:, 5])); np.add.at(cnt, owner, 1)
    out = {}
    for f, s, floor in (("coll_fx", sF, fl["force"]), ("coll_fy", sF, fl["force"]), ("coll_trq", sT, fl["torque"])):
        out[f] = assert_elementwise(f, hw.get(f), ow.get(f), rtol, floor * np.maximum(cnt, 1), scale=s)
    return out


def compare_worlds(hw, ow, rtol=1e-10, fields=SCALARS, check_pairs=True, check_inter=True):
    assert hw.M == ow.M
    out = {}
    if check_pairs:
        out["n_pairs"] = compare_pairs(hw, ow)
    hid, hg, hs = hw.ids(); oid, og, os_ = ow.ids()
    assert np.array_equal(hid, oid) and np.array_equal(hg, og)
    assert np.array_equal(hs, os_), "status tags differ"
    if check_inter:
        out.update(compare_interactions(hw, ow, rtol))
    if any(f in TOTALS for f in fields):
        out.update(compare_totals(hw, ow, rtol))
    for f in fields:
        if f in TOTALS:
            continue
        a, b = hw.get(f), ow.get(f)
        if f[:2] in ("sa", "si") and len(f) == 4:
            # a component of a stress tensor on the scale of the tensor: the resident steps sum the rows' products exactly (fixed-point totals),
            # the oracle in the reference's serial order -- where a component cancels analytically (two stars meeting head-on: s22) one leaves
            # 0, the other 1e-29 of round-off beside components of 60
            scale = max(max(np.abs(ow.get(f[:2] + c)).max() for c in ("11", "12", "21", "22")), 1e-300)
            e = float(np.max(np.abs(a - b)) / scale)
        elif f in ("e12", "e21"):
            # the shear strain of a rigid rotation cancels analytically (update_floe.jl:436-446):
            # what is stored is round-off, so compare it on the scale of the normal components
            scale = max(np.abs(ow.get("e11")).max(), np.abs(ow.get("e22")).max(), 1e-300)
            e = float(np.max(np.abs(a - b)) / scale)
        else:
            e = relerr(a, b)
        assert e <= rtol, (f, e)
        out[f] = e
    ho, hx, hy = hw.rings(); oo, ox, oy = ow.rings()
    assert np.array_equal(ho, oo)
    e = max(relerr(hx, ox), relerr(hy, oy))
    assert e <= rtol, ("vertices", e)
    out["vertices"] = e
    return out
