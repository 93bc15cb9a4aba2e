"""Output path on the resident state: the GridOutputWriter averages
(calc_eulerian_data!, output.jl:793-914) and the "nothing to simplify" test (simplification.jl:339-378).
The reference holds no numeric fixture for either (test/test_output.jl checks names and shapes): the oracle is
checked on analytic cases and conservation, the HIP path against the oracle."""
import numpy as np
import pytest

import cases
import parity

# the one floe of the reference's output test (test/test_output.jl:12-13) on its 10 x 5 grid (:28-35)
REF_FLOE = [[7.5e4, 7.5e4], [7.5e4, 9.5e4], [9.5e4, 9.5e4], [9.5e4, 7.5e4], [7.5e4, 7.5e4]]
REF_XG = np.linspace(-1e5, 1e5, 11)
REF_YG = np.linspace(-1e5, 1e5, 6)


def mk():
    import subzero_jl_amd
    return subzero_jl_amd.World(0)


def omk():
    from oracle import orc
    return orc.World()


def _names():
    from oracle import orc
    return orc.EUL_OUTPUTS


def _reference_case(make):
    w = make()
    w.set_consts(); w.set_settings()
    w.set_domain([0, 0, 0, 0], -1e5, 1e5, -1e5, 1e5)
    w.add_floe(np.array(REF_FLOE), 0.5)
    w.set("u", [0.3]); w.set("v", [-0.1]); w.set("p_dudt", [1e-3]); w.set("p_dvdt", [-2e-3]); w.set("overarea", [7.0])
    w.set("sa11", [3.0]); w.set("sa12", [1.0]); w.set("sa21", [1.0]); w.set("sa22", [-2.0])
    w.set("e11", [1e-3]); w.set("e12", [2e-3]); w.set("e21", [2e-3]); w.set("e22", [4e-3])
    return w


def _check_reference_case(w, d):
    k = _names().index
    area = np.zeros((10, 5)); area[8, 4] = 0.5e4 * 2e4; area[9, 4] = 1.5e4 * 2e4
    assert parity.relerr(d[k("area_grid")], area) < 1e-13
    assert parity.relerr(d[k("si_frac_grid")], area / (2e4 * 4e4)) < 1e-13
    mass = w.get("mass")[0]
    assert parity.relerr(d[k("mass_grid")], mass * area / 4e8) < 1e-13
    occ = (area > 0).astype(float)
    for name, val in (("u_grid", 0.3), ("v_grid", -0.1), ("dudt_grid", 1e-3), ("dvdt_grid", -2e-3), ("overarea_grid", 7.0),
                      ("height_grid", 0.5), ("stress_xx_grid", 3.0), ("stress_yx_grid", 1.0), ("stress_xy_grid", 1.0),
                      ("stress_yy_grid", -2.0), ("strain_ux_grid", 1e-3), ("strain_vx_grid", 2e-3), ("strain_uy_grid", 2e-3),
                      ("strain_vy_grid", 4e-3)):
        assert parity.relerr(d[k(name)], val * occ) < 1e-13, name
    # largest eigenvalue of [[3, 1], [1, -2]]
    assert parity.relerr(d[k("stress_eig_grid")], (0.5 + np.sqrt(6.25 + 1.0)) * occ) < 1e-13


def test_oracle_eulerian_reference_floe():
    w = _reference_case(omk)
    _check_reference_case(w, w.eulerian_data(REF_XG, REF_YG))


def test_oracle_eulerian_conservation():
    """the grid covers every floe, so the cells' area and mass add up to the floes' (topography: minus the part it covers)"""
    from subzero_jl_amd import fields
    for topo in (False, True):
        cfg = fields.make_config(n_floes=150, seed=4, walls=True, topography=topo, ocean="strait")
        w = fields.build_world(omk(), cfg)
        L = cfg["L"]
        d = w.eulerian_data(np.linspace(-0.125 * L, 1.125 * L, 11), np.linspace(-0.1 * L, 1.1 * L, 7))
        k = _names().index
        if not topo:
            assert abs(d[k("area_grid")].sum() / w.get("area").sum() - 1) < 1e-12
            assert abs(d[k("mass_grid")].sum() / w.get("mass").sum() - 1) < 1e-12
        else:
            assert 0.5 < d[k("area_grid")].sum() / w.get("area").sum() <= 1 + 1e-12
        assert np.all(d[k("si_frac_grid")] <= 1 + 1e-12) and d[k("si_frac_grid")].max() > 0.3


def test_oracle_simplify_check():
    w = _reference_case(omk)
    assert list(w.simplify_check(30, 1e6, 0.1)) == [0, 0, 0, 0]
    assert list(w.simplify_check(4, 1e6, 0.1)) == [0, 0, 1, 0]          # 5 ring points > 4
    assert list(w.simplify_check(30, 1e9, 0.1)) == [0, 0, 0, 1]         # 4e8 m^2 < 1e9
    assert list(w.simplify_check(30, 1e6, 0.6)) == [0, 0, 0, 1]         # 0.5 m < 0.6
    w.set_status(np.array([cases.REMOVE], np.int32))
    assert list(w.simplify_check(30, 1e9, 0.6)) == [1, 0, 0, 0]         # removed floes are not dissolved again


# ---------------------------------------------------------------- HIP path
@pytest.mark.gpu
def test_eulerian_reference_floe():
    w = _reference_case(mk)
    _check_reference_case(w, w.eulerian_data(REF_XG, REF_YG))
    sub = w.eulerian_data(REF_XG, REF_YG, ["mass_grid", "u_grid"])
    full = w.eulerian_data(REF_XG, REF_YG)
    assert np.array_equal(sub[0], full[5]) and np.array_equal(sub[1], full[0])
    with pytest.raises(Exception):
        w.eulerian_data(REF_XG, REF_YG, ["no_such_grid"])
    with pytest.raises(Exception):
        w.eulerian_data(np.array([0.0, 1.0, 3.0]), REF_YG)


@pytest.mark.gpu
@pytest.mark.parametrize("walls,topo,dims", [(False, False, (10, 7)), (False, False, (64, 48)), (True, True, (12, 12)), (True, True, (50, 40))])
def test_eulerian_random(walls, topo, dims):
    """all 18 grid outputs against the oracle after a few steps (contacts, rotation, stress history), with the
    ghosts in the list as write_data! sees them (simulation.jl:102-105); walled case: topography out of the cells"""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=600, seed=21, walls=walls, topography=topo, ocean="strait" if walls else "converge_diverge")
    hw, ow = fields.build_world(mk(), cfg), fields.build_world(omk(), cfg)
    hw.run(4, 0, cfg["dt"], coupling_dt=2, stop_on_tags=False)      # (the walled field tags a floe in step 0; the oracle does not simplify either)
    for t in range(4):
        ow.timestep_sim(t, cfg["dt"], coupling_dt=2)
    L = cfg["L"]
    xg, yg = np.linspace(0, L, dims[0] + 1), np.linspace(0, L, dims[1] + 1)
    n0 = ow.M
    hw.add_ghosts(); ow.add_ghosts()
    assert hw.M == ow.M and (walls or hw.M > n0)
    got, ref = hw.eulerian_data(xg, yg), ow.eulerian_data(xg, yg)
    hw.remove_ghosts(); ow.remove_ghosts(n0)
    names = _names()
    assert np.count_nonzero(ref[names.index("area_grid")]) > 0.5 * dims[0] * dims[1]
    for k, n in enumerate(names):
        assert np.abs(ref[k]).max() > 0, n
        # the off-diagonal strain of a rigid motion is round-off of the diagonal's size: one scale per tensor
        grp = [j for j, o in enumerate(names) if o.split("_")[0] == n.split("_")[0]] if n.startswith(("stress", "strain")) else [k]
        scale = max(np.abs(ref[j]).max() for j in grp)
        assert np.abs(got[k] - ref[k]).max() < 1e-9 * scale, n
    if topo:          # some cell really lost area to the topography
        cell = (xg[1] - xg[0]) * (yg[1] - yg[0])
        frac = ref[names.index("si_frac_grid")]; area = ref[names.index("area_grid")]
        m = area > 0
        assert np.any(area[m] / frac[m] < 0.999 * cell)
    # write_grid_data: the same numbers with the ghost bracket done inside
    assert parity.relerr(hw.write_grid_data(xg, yg), ref) < 1e-9
    assert hw.M == n0


@pytest.mark.gpu
def test_simplify_check():
    from subzero_jl_amd import fields
    cfg = cases.floe_onto_island(fields.make_config(n_floes=900, seed=3, walls=True, topography=True, ocean="strait"))
    hw, ow = fields.build_world(mk(), cfg), fields.build_world(omk(), cfg)
    hw.timestep_collisions(900, cfg["dt"]); ow.timestep_collisions(900, cfg["dt"])
    area = ow.get("area")
    for args in ((30, 1e6, 0.1), (12, float(np.median(area)), 0.1), (10, 1e6, 0.3)):
        ref = [int(v) for v in ow.simplify_check(*args)]
        got = hw.simplify_check(*args)
        assert [got["remove"], got["fuse"], got["over_max_vertices"], got["dissolve"]] == ref, args
    assert ref[0] > 0 and ref[2] > 0
