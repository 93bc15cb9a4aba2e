"""Parity helpers: compare the HIP engine's state with the CPU oracle's on the same inputs."""
import numpy as np

SCALARS = ["cx", "cy", "alpha", "u", "v", "xi", "height", "mass", "moment", "p_dxdt", "p_dydt", "p_dalphadt",
           "p_dudt", "p_dvdt", "p_dxidt", "fxOA", "fyOA", "trqOA", "hflx_factor", "overarea",
           "coll_fx", "coll_fy", "coll_trq",
           "sa11", "sa12", "sa21", "sa22", "si11", "si12", "si21", "si22", "e11", "e12", "e21", "e22"]


def relerr(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    if a.size == 0 and b.size == 0:
        return 0.0
    scale = max(np.max(np.abs(b)), 1e-300)
    return float(np.max(np.abs(a - b)) / scale)


def compare_pairs(hw, ow):
    hi, hj = hw.pairs(); oi, oj = ow.pairs()
    assert len(hi) == len(oi), (len(hi), len(oi))
    assert np.array_equal(hi, oi) and np.array_equal(hj, oj), "overlap-pair indices differ"
    return len(hi)


def compare_interactions(hw, ow, rtol):
    hoff, hrows = hw.interactions(); ooff, orows = ow.interactions()
    n = min(len(hoff), len(ooff))
    assert np.array_equal(hoff[:n], ooff[:n]), "interaction row counts differ"
    m = hoff[n - 1]
    hrows, orows = hrows[:m], orows[:m]
    assert np.array_equal(hrows[:, 0], orows[:, 0]), "partner indices differ"
    errs = {}
    for c, name in ((1, "xforce"), (2, "yforce"), (5, "torque"), (6, "overlap")):
        errs[name] = relerr(hrows[:, c], orows[:, c])
    # force points are coordinates: compare against the coordinate scale
    errs["point"] = max(relerr(hrows[:, 3], orows[:, 3]), relerr(hrows[:, 4], orows[:, 4]))
    for k, v in errs.items():
        assert v <= rtol, (k, v)
    return errs


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
    for f in fields:
        a, b = hw.get(f), ow.get(f)
        if f in ("e12", "e21"):
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
