"""Parity of the HIP engine (through the C-ABI) with the CPU oracle and with the reference's
own known-answer tests.  Needs a real MI355X: run with -m gpu."""
import numpy as np
import pytest

import cases
import parity

pytestmark = pytest.mark.gpu


def mk():
    import subzero_jl_amd
    return subzero_jl_amd.World(0)


def omk():
    from oracle import orc
    return orc.World()


def test_stress_strain(golden):
    """calc_stress! / calc_strain! on the two floes of the reference's stress_strain.jld2 fixture"""
    U = golden["update_floe"]
    w = cases.run_stress_strain(mk, U)
    cases.check_stress_strain(w, U)
    o = cases.run_stress_strain(omk, U)
    for k in ("si11", "si12", "si22", "sa11", "sa22", "e11", "e22"):
        assert parity.relerr(w.get(k), o.get(k)) < 1e-12, k


def test_two_way_coupling_analytic():
    cases.check_two_way_analytic(*cases.run_two_way_analytic(mk))


@pytest.mark.parametrize("walls", [False, True])
def test_two_way_coupling_random(walls):
    """calc_two_way_coupling! on a random field against the oracle: stress on the ocean, sea-ice fraction and
    heat-flux factor per centre cell; periodic (floes shifted across the domain) and walled (trimmed cells)."""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=400, seed=11, walls=walls, ocean="converge_diverge")
    hw, ow = _pair(cfg)
    Nx, Ny = cfg["Nx"], cfg["Ny"]
    rng = np.random.default_rng(5)
    tocn = rng.uniform(-2, 2, (Nx + 1, Ny + 1)); tatm = rng.uniform(-20, 0, (Nx + 1, Ny + 1))
    for w in (hw, ow):
        w.set_two_way(True, dt=cfg["dt"]); w.set_temps(tocn, tatm)
    # a few steps first so that rotated floes, wrapped parents and contacts are in play
    hw.run(3, 0, cfg["dt"], coupling_dt=1)
    for t in range(3):
        ow.timestep_sim(t, cfg["dt"], coupling_dt=1)
    got = hw.ocean_stress(); ref = ow.ocean_stress()
    assert np.count_nonzero(ref[2]) > 1000
    for name, g, r in zip(("tau_x", "tau_y", "si_frac", "hflx"), got, ref):
        assert parity.relerr(g, r) < 1e-9, name
    parity.compare_worlds(hw, ow, rtol=1e-9, fields=["fxOA", "fyOA", "trqOA", "hflx_factor"])


def test_mixed_precision_forcings():
    """BASELINE configs[4]: mixed precision.  The reference has no Float32 answers (documentation.md:25), so the
    mixed path (per-point forcing arithmetic in fp32, totals fp64) is held to the fp64 path: forcings to 1e-5 of
    the field's largest value, 10-step trajectories to 1e-6 on velocities."""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=2000, seed=13, concentration=0.25, ocean="converge_diverge")
    h64 = fields.build_world(mk(), cfg); h32 = fields.build_world(mk(), cfg)
    h32.set_precision("mixed")
    h64.timestep_coupling(); h32.timestep_coupling()
    for k in ("fxOA", "fyOA", "trqOA", "hflx_factor"):
        a, b = h32.get(k), h64.get(k)
        assert np.max(np.abs(a - b)) <= 1e-5 * max(np.max(np.abs(b)), 1e-300), k
    assert not np.array_equal(h32.get("fxOA"), h64.get("fxOA"))        # it really is the other kernel
    h64.run(10, 0, cfg["dt"], coupling_dt=1); h32.run(10, 0, cfg["dt"], coupling_dt=1)
    for k in ("u", "v", "xi"):
        a, b = h32.get(k), h64.get(k)
        assert np.max(np.abs(a - b)) <= 1e-6 * np.max(np.abs(b)), k
    assert np.max(np.abs(h32.get("cx") - h64.get("cx"))) < 1e-3        # metres, after 200 s


@pytest.mark.parametrize("shape,rtol", [("star", 1e-9), ("voronoi", 1e-6)])
def test_config0_shear_flow_two_way(shape, rtol):
    """BASELINE configs[0] (examples/shear_flow.jl): ~100 floes in a 100 km doubly periodic box, shear ocean,
    collisions on, two-way coupling on; 20 timesteps against the oracle -- with star polygons and with the reference's own
    kind of field, Voronoi cells that touch (criterion 1e-6 there: test_voronoi_field_touching_cells says why)."""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=100, seed=21, spacing=1.0e4, ocean="shear", shape=shape)
    hw, ow = _pair(cfg)
    for w in (hw, ow):
        w.set_two_way(True, dt=cfg["dt"]); w.set_temps(0.0, 0.0)
    hw.run(20, 0, cfg["dt"], coupling_dt=10, stop_on_tags=False)
    for t in range(20):
        ow.timestep_sim(t, cfg["dt"], coupling_dt=10)
    parity.compare_worlds(hw, ow, rtol=rtol, check_inter=shape == "star")
    for name, g, r in zip(("tau_x", "tau_y", "si_frac"), hw.ocean_stress(), ow.ocean_stress()):
        assert parity.relerr(g, r) < rtol, name
    assert 0.6 < ow.ocean_stress()[2].mean() < 0.9       # sea-ice fraction of the 0.8-concentration field


# ---------------------------------------------------------------- the reference's known answers, through the C-ABI
@pytest.mark.parametrize("k", range(5))
def test_floe_floe(golden, k):
    G = golden["collisions"]; case = G["floe_floe"][k]
    cases.check_floe_floe(cases.run_floe_floe(mk, G, case), case)


@pytest.mark.parametrize("k", range(8))
def test_floe_boundary(golden, k):
    G = golden["collisions"]; case = G["boundary"]["cases"][k]
    cases.check_boundary(cases.run_boundary(mk, G, case), case)


@pytest.mark.parametrize("k", range(4))
def test_add_ghosts(golden, k):
    G = golden["collisions"]; case = G["add_ghosts"]["cases"][k]
    cases.check_add_ghosts(cases.run_add_ghosts(mk, G, case), G, case)


def test_ghost_collisions(golden):
    G = golden["collisions"]
    cases.check_ghost_collisions(cases.ghost_collision_scenarios(mk, G), exact=True)


@pytest.mark.parametrize("k", range(6))
def test_forcings(golden, k):
    F = golden["forcings"]; case = F["cases"][k]
    cases.check_forcing(cases.run_forcing(mk, F, case), case)


@pytest.mark.parametrize("ns,ew", [("open", "open"), ("periodic", "open"), ("open", "periodic"), ("periodic", "periodic")])
def test_in_bounds_and_lattice_sample(golden, ns, ew):
    """The forcing kernels' own in-bounds test and lattice sample (sz_debug_sample_fields) against the reference's vectors: the four in_bounds
    truth tables (test_coupling.jl:181-197; coupling.jl:494-597), and find_interp_knots (test_coupling.jl:199-282; coupling.jl:702-797) -- which
    grid lines a bilinear blend reads at a point, beyond each edge too: in a periodic direction line ncells + 1 is line 1 and the window wraps;
    in a non-periodic one it ends at the grid's edge.  A lattice with a distinct value per line makes the lines visible in the sampled value
    (held to the oracle's sample, 1e-12)."""
    from oracle import orc as O
    G = golden["coupling_grid"]; g = G["grid"]; B = G["in_bounds"]
    kind = {"open": 0, "periodic": 1}
    kinds = [kind[ns], kind[ns], kind[ew], kind[ew]]
    Nx = int(round((g["xf"] - g["x0"]) / g["dx"])); Ny = int(round((g["yf"] - g["y0"]) / g["dy"]))
    ix, iy = np.meshgrid(np.arange(Nx + 1, dtype=float), np.arange(Ny + 1, dtype=float), indexing="ij")
    uo, vo = ix.copy(), iy.copy()                      # uocn: the x line's number - 1, vocn: the y line's
    if ew == "periodic":
        uo[Nx, :] = uo[0, :]                            # (a periodic ocean: the last line is the first)
    if ns == "periodic":
        vo[:, Ny] = vo[:, 0]
    hf = 100.0 * uo + vo; ua = -uo; va = 0.5 * vo
    hw, ow = mk(), omk()
    for w in (hw, ow):
        w.set_domain(kinds, g["x0"], g["xf"], g["y0"], g["yf"])
        w.set_grid_fields(Nx, Ny, g["x0"], g["xf"], g["y0"], g["yf"], uo, vo, hf, ua, va)
    per_x, per_y = ew == "periodic", ns == "periodic"
    # in_bounds: the reference's table for this pair of kinds
    got = hw.sample_fields(B["x"], B["y"])
    assert [bool(v) for v in got[:, 0]] == B[f"{ns}_{ew}"]
    # find_interp_knots through the sample: points beyond each edge (they are only sampled where in_bounds admits them), on lines, inside cells
    xs = np.array([-13.0, -10.5, -10.0, -9.5, -7.0, 0.0, 3.25, 9.5, 10.0, 10.5, 13.0]); ys = np.array([-11.0, -8.0, -7.0, -1.0, 3.0, 7.5, 8.0, 9.0, 13.0])
    X, Y = [a.ravel() for a in np.meshgrid(xs, ys, indexing="ij")]
    got = hw.sample_fields(X, Y)
    n_checked = 0
    for k, (x, y) in enumerate(zip(X, Y)):
        assert bool(got[k, 0]) == ow.in_bounds(x, y, per_x, per_y)
        if not got[k, 0]:
            continue
        lines, (tx, ty) = ow.sample_lines(x, y, per_x, per_y)
        want = ow.sample_fields(x, y, per_x, per_y)
        assert np.allclose(got[k, 1:6], want, rtol=0, atol=1e-12 * 1e3), (x, y, got[k], want)
        # the lines, up to the one freedom of a point ON a grid line (either neighbouring cell gives the same value)
        on_x, on_y = abs((x - g["x0"]) / g["dx"] - round((x - g["x0"]) / g["dx"])) < 1e-12, abs((y - g["y0"]) / g["dy"] - round((y - g["y0"]) / g["dy"])) < 1e-12
        if not on_x:
            assert [int(got[k, 6]), int(got[k, 7])] == lines[:2] and abs(got[k, 10] - tx) < 1e-12, (x, y, got[k], lines)
        if not on_y:
            assert [int(got[k, 8]), int(got[k, 9])] == lines[2:] and abs(got[k, 11] - ty) < 1e-12, (x, y, got[k], lines)
        n_checked += 1
    assert n_checked >= (len(X) if per_x and per_y else 20)
    # and the reference's own knot windows (glines 0:10:80): the two lines read inside each knot interval are its knot_idx
    K = G["knots_grid"]
    for c in G["find_interp_knots"]:
        if c["periodic"] != per_x:
            continue
        w = mk()
        w.set_domain(kinds, K["g0"], K["g0"] + K["L"], 0.0, 40.0)
        n = K["ncells"]
        line = np.arange(n + 1, dtype=float)[:, None] * np.ones((1, 5))
        if per_x:
            line[n] = line[0]
        w.set_grid_fields(n, 4, K["g0"], K["g0"] + K["L"], 0.0, 40.0, line, 0.0, 0.0, 0.0, 0.0)
        px = [a + t * (b - a) for a, b in zip(c["knots"][:-1], c["knots"][1:]) for t in (0.25, 0.5, 0.875)]
        r = w.sample_fields(px, [15.0] * len(px))
        q = 0
        for (a, b, ia, ib) in zip(c["knots"][:-1], c["knots"][1:], c["idx"][:-1], c["idx"][1:]):
            for t in (0.25, 0.5, 0.875):
                assert r[q, 0] == 1.0 and (int(r[q, 6]), int(r[q, 7])) == (ia, ib) and abs(r[q, 10] - t) < 1e-12, (c, px[q], r[q])
                assert abs(r[q, 1] - ((1 - t) * (ia - 1) + t * (ib - 1 if ib != 1 or not per_x else 0))) < 1e-12
                q += 1


# ---------------------------------------------------------------- seeded random fields vs the oracle
def _pair(cfg):
    from subzero_jl_amd import fields
    return fields.build_world(mk(), cfg), fields.build_world(omk(), cfg)


@pytest.mark.parametrize("n,seed", [(300, 1), (2000, 2)])
def test_collisions_random_periodic(n, seed):
    """add_ghosts! + timestep_collisions!: pair list bit-exact, forces within 1e-10 relative."""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=n, seed=seed)
    hw, ow = _pair(cfg)
    hw.add_ghosts(); ow.add_ghosts()
    assert hw.M == ow.M
    assert hw.ghosts() == ow.ghosts()
    hw.timestep_collisions(n, cfg["dt"]); ow.timestep_collisions(n, cfg["dt"])
    res = parity.compare_worlds(hw, ow, rtol=1e-10, fields=["coll_fx", "coll_fy", "coll_trq", "overarea"])
    assert res["n_pairs"] > n          # the field really is in contact
    assert ow.interactions()[0][-1] > n // 2


def test_collisions_walls_topography():
    """config-4 style: four collision walls + the strait of examples/simple_strait.jl (two coast wedges and an island):
    the floe-boundary and floe-topography clip paths.  One floe is put on the island: it is tagged for removal
    (collisions.jl:525)."""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=900, seed=3, walls=True, topography=True, ocean="strait")
    assert len(cfg["topography"]) == 3
    cases.floe_onto_island(cfg)
    hw, ow = _pair(cfg)
    hw.timestep_collisions(900, cfg["dt"]); ow.timestep_collisions(900, cfg["dt"])
    parity.compare_worlds(hw, ow, rtol=1e-10, fields=["coll_fx", "coll_fy", "coll_trq", "overarea"])
    rows = ow.interactions()[1]
    assert np.sum((rows[:, 0] < 0) & (rows[:, 0] >= -4)) > 10 and np.sum(rows[:, 0] < -4) > 10      # wall and topography contacts
    # the tag counts a host uses to skip simplify_floes! when nothing was removed or fused
    st = hw.stats(); tags = hw.ids()[2]
    assert st["n_status_remove"] == int((tags == cases.REMOVE).sum()) and st["n_status_fuse"] == int((tags == cases.FUSE).sum())
    assert st["n_status_remove"] == int((ow.ids()[2] == cases.REMOVE).sum()) > 0


def test_moving_boundary_compression():
    """MovingBoundary walls (examples/moving_bounds.jl style): the north and south walls close in,
    contribute their velocity to the friction and are moved after the contacts (collisions.jl:797)."""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=400, seed=8, walls=True, ocean="strait")
    L = cfg["L"]
    worlds = []
    for w in (mk(), omk()):
        w.set_consts(E=cfg["E"]); w.set_settings()
        w.set_domain([3, 3, 2, 2], 0.0, L, 0.0, L, bu=[0.05, -0.05, 0, 0], bv=[-0.4, 0.4, 0, 0])   # N,S moving; E,W collision
        w.set_grid_fields(cfg["Nx"], cfg["Ny"], 0.0, L, 0.0, L, cfg["uo"], cfg["vo"], cfg["hf"], cfg["ua"], cfg["va"])
        off, vx, vy = cfg["vert_off"], cfg["vx"], cfg["vy"]
        if hasattr(w, "load_columns"):
            d = cfg["derived"]
            w.load_columns(dict(cx=d["cx"], cy=d["cy"], rmax=d["rmax"], area=d["area"], height=d["height"], mass=d["mass"],
                                moment=d["moment"], u=cfg["u"], v=cfg["v"], xi=cfg["xi"], vert_off=off, vx=vx, vy=vy))
            w.set_subpoints_csr(cfg["sub_off"], cfg["sx"], cfg["sy"])
        else:
            so = cfg["sub_off"]
            for i in range(cfg["n_floes"]):
                w.add_floe(np.stack([vx[off[i]:off[i + 1]], vy[off[i]:off[i + 1]]], 1), cfg["height"][i])
                w.set_subpoints(i, cfg["sx"][so[i]:so[i + 1]], cfg["sy"][so[i]:so[i + 1]])
            w.set("u", cfg["u"]); w.set("v", cfg["v"]); w.set("xi", cfg["xi"])
        worlds.append(w)
    hw, ow = worlds
    for t in range(6):
        hw.timestep_sim(t, cfg["dt"], coupling_dt=1); ow.timestep_sim(t, cfg["dt"], coupling_dt=1)
    assert np.array_equal(hw.boundary_vals(), ow.boundary_vals())
    assert hw.boundary_vals()[0] == L - 0.4 * cfg["dt"] * 6
    parity.compare_worlds(hw, ow, rtol=1e-9, check_pairs=True)
    assert np.any(ow.interactions()[1][:, 0] < 0)


def test_forcing_and_integrator_random():
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=500, seed=4, ocean="converge_diverge")
    hw, ow = _pair(cfg)
    hw.timestep_coupling(); ow.timestep_coupling()
    for f in ("fxOA", "fyOA", "trqOA", "hflx_factor"):
        assert parity.relerr(hw.get(f), ow.get(f)) <= 1e-11, f
    hw.timestep_floe_properties(cfg["dt"]); ow.timestep_floe_properties(cfg["dt"])
    parity.compare_worlds(hw, ow, rtol=1e-11, check_pairs=False, check_inter=False)


def test_blocked_subfloe_points_are_a_reordering(monkeypatch):
    """The one-way forcing loop reads a blocked copy of the sub-floe points (State::sxy, Morton order per floe): the same points, another
    order of the per-floe sums -- forcings within 1e-12 of the loop on the caller's order (SZ_BLOCK_POINTS=0), both within 1e-11 of the oracle
    (coupling.jl:1486-1589), and the points handed back to the host are the caller's, in the caller's order."""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=700, seed=14, ocean="converge_diverge")
    ow = fields.build_world(omk(), cfg)
    ow.timestep_coupling()
    got = {}
    for env in ("1", "0"):
        monkeypatch.setenv("SZ_BLOCK_POINTS", env)
        hw = fields.build_world(mk(), cfg)
        off0, sx0, sy0 = hw.subpoints()
        hw.timestep_coupling()
        hw.run(2, 0, cfg["dt"], coupling_dt=1)          # (resident steps: the forcings in the tail of the narrow launch read the same copy)
        off1, sx1, sy1 = hw.subpoints()
        assert np.array_equal(off0, off1) and np.array_equal(sx0, sx1) and np.array_equal(sy0, sy1)
        hw2 = fields.build_world(mk(), cfg)
        hw2.timestep_coupling()
        got[env] = {f: hw2.get(f).copy() for f in ("fxOA", "fyOA", "trqOA", "hflx_factor")}
        for f in got[env]:
            assert parity.relerr(got[env][f], ow.get(f)) <= 1e-11, (env, f)
    for f in got["1"]:
        assert parity.relerr(got["1"][f], got["0"][f]) <= 1e-12, f
    assert not np.array_equal(got["1"]["fxOA"], got["0"]["fxOA"])          # (it really is another order)


@pytest.mark.parametrize("n,seed,steps", [(400, 5, 10), (2500, 6, 4)])
def test_trajectories(n, seed, steps):
    """timestep_sim! for several steps, state resident on the device: trajectories within 1e-9."""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=n, seed=seed)
    hw, ow = _pair(cfg)
    assert hw.run(steps, 0, cfg["dt"], coupling_dt=1) == steps        # no floe is tagged: the batch runs through
    for t in range(steps):
        ow.timestep_sim(t, cfg["dt"], coupling_dt=1)
    parity.compare_worlds(hw, ow, rtol=1e-9)
    assert np.array_equal(hw.warn_counts(), ow.warn_counts())


@pytest.mark.parametrize("n,seed,walls,conc,steps", [(150, 3, False, 0.7, 30), (400, 4, True, 0.85, 30), (1500, 5, False, 1.0, 12)])
def test_voronoi_field_touching_cells(n, seed, walls, conc, steps):
    """The reference draws its floe fields from a Voronoi tessellation (initialize_floe_field; BASELINE configs[0]): convex cells
    with 3-9 vertices that TOUCH their neighbours along whole edges and the walls along whole sides -- every contact starts as a
    degenerate one (collinear edges, shared vertices, zero overlap area) and becomes a sliver as the shear flow moves the cells.
    First call: pair lists bit-exact, rows within 1e-10.  Then 30 resident steps (12 of the fully packed, jammed field): pair lists equal, guard counters equal, state columns
    within 1e-6 -- not the 1e-9 of the star-polygon fields: the contact point of a sliver region is its centroid, conditioned like
    coordinate^2 / area, and ONE such contact turns a last-bit difference of a rotation angle (the forcing sums of the two codes differ
    in the order of their additions, as the tolerance allows) into 1e-9 of a torque in one step (step 2 of the fully packed case, step 16
    of the walled one: `tools/voronoi_diverge.py` prints the history).  The reference's own trajectory is as sensitive to its last bits."""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=n, seed=seed, spacing=1.0e4, shape="voronoi", ocean="shear", concentration=conc, walls=walls)
    assert 4 <= np.diff(cfg["vert_off"]).min() and np.diff(cfg["vert_off"]).max() <= 14
    hw, ow = _pair(cfg)
    for w in (hw, ow):
        w.add_ghosts(); w.timestep_collisions(n, cfg["dt"])
    assert hw.M == ow.M and hw.ghosts() == ow.ghosts()
    parity.compare_pairs(hw, ow)
    parity.compare_interactions(hw, ow, rtol=1e-10)
    for w in (hw, ow):
        w.remove_ghosts(n)
    hw2, ow2 = _pair(cfg)
    hw2.run(steps, 0, cfg["dt"], coupling_dt=5, stop_on_tags=False)
    for t in range(steps):
        ow2.timestep_sim(t, cfg["dt"], coupling_dt=5)
    parity.compare_worlds(hw2, ow2, rtol=1e-6, check_inter=False)
    parity.compare_pairs(hw2, ow2)
    assert np.array_equal(hw2.warn_counts(), ow2.warn_counts())
    assert np.count_nonzero(ow2.get("overarea")) > n // 4            # the cells did run into each other


@pytest.mark.parametrize("walls", [False, True])
def test_resident_and_process_mode_interleaved(walls):
    """Resident steps keep a fixed broad-phase grid and its cell lists across steps; process-mode calls fit their
    own grid and move floes without re-binning.  Interleaving both (and a host-side edit in between) must give
    the oracle's trajectory."""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=600, seed=17, walls=walls, topography=walls)
    hw, ow = _pair(cfg)
    dt = cfg["dt"]
    t = 0

    def both_resident(k):
        nonlocal t
        # (the walled field has a floe removed by the topography in step 0; the oracle does not simplify either)
        hw.run(k, t, dt, coupling_dt=1, stop_on_tags=False)
        for q in range(k):
            ow.timestep_sim(t + q, dt, coupling_dt=1)
        t += k

    def both_process():
        nonlocal t
        for w in (hw, ow):
            n = w.M
            w.add_ghosts(); w.timestep_collisions(n, dt); w.remove_ghosts(n)
            w.timestep_coupling(); w.timestep_floe_properties(dt)
        t += 1

    both_resident(3)
    both_process()
    both_resident(2)
    for w in (hw, ow):                      # host-side edit: forces a new upload
        u = w.get("u"); u[::7] += 0.05; w.set("u", u)
    both_resident(2)
    both_process()
    both_resident(3)
    parity.compare_worlds(hw, ow, rtol=1e-9)
    parity.compare_pairs(hw, ow)


@pytest.mark.parametrize("seed,periodic", [(500, True), (505, False)])
def test_rough_shapes(seed, periodic):
    """3 to 40 vertices, deep radial variation (concave, spiky), heavy overlaps: several contact regions per pair,
    rings above the 8-lane kernels' capacity (the 16- and 64-lane variants run), working-set retries."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_parity_shapes", os.path.join(os.path.dirname(__file__), "..", "tools", "rough_shapes.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    hw = mod.rough_world(mk, seed, 12, periodic); ow = mod.rough_world(omk, seed, 12, periodic)
    for w in (hw, ow):
        n = w.M
        w.add_ghosts(); w.timestep_collisions(n, 10)
    assert parity.compare_pairs(hw, ow) > 300
    parity.compare_interactions(hw, ow, 1e-9)
    parity.compare_worlds(hw, ow, rtol=1e-9, fields=["coll_fx", "coll_fy", "coll_trq", "overarea"])
    assert max(len(r) for r in (hw.ring(i) for i in range(0, 144, 7))) > 20


def test_narrow_variant_retry_many_crossings():
    """two 8-spike stars crossing 16 times: more crossings than the small narrow-phase working set
    holds, so the item is redone by the largest variant; rows must still match the oracle."""
    th = np.arange(16) * (2 * np.pi / 16)
    rad = np.where(np.arange(16) % 2 == 0, 1.0e4, 0.55e4)
    star = lambda rot, cx: np.stack([cx + rad * np.cos(-th + rot), 3e4 + rad * np.sin(-th + rot)], 1)
    res = []
    for w in (mk(), omk()):
        w.set_domain([0, 0, 0, 0], -1e5, 1e5, -1e5, 1e5)
        w.add_floe(cases.closed(star(0.0, 1.0e4)), 0.25)
        w.add_floe(cases.closed(star(np.pi / 8, 1.05e4)), 0.25)
        cases.set_vel(w, 0, {"u": 0.1}); cases.set_vel(w, 1, {"v": -0.1})
        w.floe_floe_interaction(0, 1, 10, 1.0)
        res.append(w.inter(0))
    h, o = res
    assert len(o) >= 1 and h.shape == o.shape
    assert np.array_equal(h[:, 0], o[:, 0])
    fl = parity.force_floors(o, 1.0e4)
    for c, floor in ((1, fl["force"]), (2, fl["force"]), (3, 1e-10 * fl["Lc"]), (4, 1e-10 * fl["Lc"]), (6, fl["area"])):
        parity.assert_elementwise(f"column {c}", h[:, c], o[:, c], 1e-10, floor)
    from oracle import orc
    assert len(orc.intersection_points(cases.closed(star(0.0, 1.0e4)), cases.closed(star(np.pi / 8, 1.05e4)))) > 12


def test_sparse_field_config5_geometry():
    """config-5 geometry (25 % concentration: most broad-phase candidates are rejected) in fp64:
    pair list bit-exact, forces within 1e-10."""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=3000, seed=14, concentration=0.25)
    hw, ow = _pair(cfg)
    hw.add_ghosts(); ow.add_ghosts()
    hw.timestep_collisions(3000, cfg["dt"]); ow.timestep_collisions(3000, cfg["dt"])
    res = parity.compare_worlds(hw, ow, rtol=1e-10, fields=["coll_fx", "coll_fy", "coll_trq", "overarea"])
    assert res["n_pairs"] < 3000          # sparse: fewer pairs than floes


def test_converge_diverge_100k_properties():
    """config-3 size on one GPU (100 000 floes, converge/diverge ocean): capacities hold, internal contact
    forces cancel over the parents, no guard fires for the forcings, state stays finite over 5 steps."""
    from subzero_jl_amd import fields
    n = 100000
    cfg = _bench_cfg("configs2")
    hw = fields.build_world(mk(), cfg)
    hw.run(5, 0, cfg["dt"], coupling_dt=1)
    st = hw.stats()
    assert st["M"] == n and st["n_pairs"] > n and st["n_trace_fail"] == 0
    for f in ("cx", "cy", "u", "v", "xi", "coll_fx", "fxOA"):
        assert np.all(np.isfinite(hw.get(f))), f
    fx = hw.get("coll_fx")
    assert abs(fx.sum()) <= 1e-9 * np.abs(fx).max() * st["n_inter_rows"]
    assert np.all(hw.get("fxOA") != 0)


def test_full_size_properties():
    """BASELINE config 2 size (10k floes): size-independent checks (no oracle run at this size in
    the GPU suite): Newton's third law on the mirrored rows, parents only keep totals, momentum
    change equals the summed forces."""
    from subzero_jl_amd import fields
    n = 10000
    cfg = fields.make_config(n_floes=n, seed=12345)
    hw = fields.build_world(mk(), cfg)
    hw.add_ghosts()
    hw.timestep_collisions(n, cfg["dt"])
    off, rows = hw.interactions()
    M = hw.M
    owner = np.repeat(np.arange(M), np.diff(off))
    ff = rows[:, 0] > 0
    # every floe-floe row (k -> j) has a mirror row (j -> k) with the opposite force at the same point
    key = {}
    ids, gids, _ = hw.ids()
    cx, cy = hw.get("cx"), hw.get("cy")
    tot = np.zeros(2)
    for k in range(n):
        r = rows[off[k]:off[k + 1]]
        tot += r[:, 1:3].sum(0)
    fscale = np.abs(rows[:, 1:3]).max()
    assert np.all(np.abs(tot) <= 1e-9 * fscale * len(rows)), tot     # internal forces cancel over the parents
    assert abs(hw.get("coll_fx")[:n].sum()) <= 1e-9 * fscale * len(rows)
    assert np.all(hw.get("coll_fx")[n:] == 0)


# ---------------------------------------------------------------- the oracle at the sizes the metric is quoted on
import functools
import os


@functools.lru_cache(maxsize=2)
def _bench_cfg(workload):
    """the very fields bench.py times (same generator arguments as its workload table)"""
    from subzero_jl_amd import fields
    wl = {"configs1": dict(n_floes=10000, seed=12345), "configs3": dict(n_floes=10000, seed=12345, walls=True, topography=True, ocean="strait"),
          "configs2": dict(n_floes=100000, seed=12346, ocean="converge_diverge")}[workload]
    return fields.make_config(**wl)


def _cores():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(quota) // int(period)))
    except Exception:
        pass
    return n


@pytest.mark.parametrize("workload", ["configs1", "configs3"])
def test_full_size_oracle_parity(workload):
    """BASELINE configs[1] (10 000 floes, periodic box, uniform flow) and configs[3] (10 000 floes between four collision walls +
    the strait's topography) at the size the metric is quoted on, against the OpenMP oracle: ghosts and their order, pair list
    bit-exact, interaction rows and totals to 1e-10 per element; then 3 resident timesteps: pair list equal, state to 1e-9,
    guard counters equal."""
    from subzero_jl_amd import fields
    cfg = _bench_cfg(workload)
    n = cfg["n_floes"]
    hw, ow = _pair(cfg); ow.set_threads(_cores())
    hw.add_ghosts(); ow.add_ghosts()
    assert hw.M == ow.M and hw.ghosts() == ow.ghosts()
    if workload == "configs1":
        assert hw.M > n           # periodic: there are ghosts
    hw.timestep_collisions(n, cfg["dt"]); ow.timestep_collisions(n, cfg["dt"])
    res = parity.compare_worlds(hw, ow, rtol=1e-10, fields=["coll_fx", "coll_fy", "coll_trq", "overarea"])
    assert res["n_pairs"] > n and ow.interactions()[0][-1] > n
    if workload == "configs3":
        rows = ow.interactions()[1]
        assert np.sum((rows[:, 0] < 0) & (rows[:, 0] >= -4)) > 50 and np.sum(rows[:, 0] < -4) > 20      # wall and topography contacts
    del hw, ow
    hw, ow = _pair(cfg); ow.set_threads(_cores())
    steps = 3
    assert hw.run(steps, 0, cfg["dt"], coupling_dt=1) == steps
    for t in range(steps):
        ow.timestep_sim(t, cfg["dt"], coupling_dt=1)
    parity.compare_worlds(hw, ow, rtol=1e-9)
    assert np.array_equal(hw.warn_counts(), ow.warn_counts())


def test_collision_call_100k_pairs_bit_exact():
    """BASELINE configs[2] size on one GPU: add_ghosts! + timestep_collisions! of 100 000 floes (converge/diverge field) against the
    oracle's O(M^2) pair loop -- ghost lists equal, overlap-pair indices bit-exact, rows and totals to 1e-10 per element."""
    cfg = _bench_cfg("configs2")
    n = cfg["n_floes"]
    hw, ow = _pair(cfg); ow.set_threads(_cores())
    hw.add_ghosts(); ow.add_ghosts()
    assert hw.M == ow.M and hw.ghosts() == ow.ghosts()
    hw.timestep_collisions(n, cfg["dt"]); ow.timestep_collisions(n, cfg["dt"])
    res = parity.compare_worlds(hw, ow, rtol=1e-10, fields=["coll_fx", "coll_fy", "coll_trq", "overarea"])
    assert res["n_pairs"] > n


def test_resident_step_100k_against_the_oracle():
    """BASELINE configs[2] at full size through the RESIDENT path: four whole timesteps (ghosts, collisions, forcings of the converge / diverge
    ocean, update) of 100 000 floes in one sz_step batch -- pipelined steps, the forcings on their second stream -- against the oracle: pair
    list of the last step equal, interaction rows and totals 1e-10 per element, state 1e-9, guard counters equal.  (The collision call alone
    at this size: the test above.)"""
    import os
    cfg = _bench_cfg("configs2")
    os.environ["SZ_PIPE_MAX_FLOES"] = "1000000"          # (fields of this size take the three-launch steps by default: the pipelined ones are held to the oracle here)
    try:
        hw, ow = _pair(cfg)
    finally:
        del os.environ["SZ_PIPE_MAX_FLOES"]
    ow.set_threads(_cores())
    steps = 4
    assert hw.run(steps, 0, cfg["dt"], coupling_dt=1) == steps
    assert hw.pipelined()
    for t in range(steps):
        ow.timestep_sim(t, cfg["dt"], coupling_dt=1)
    res = parity.compare_worlds(hw, ow, rtol=1e-9)
    assert res["n_pairs"] > cfg["n_floes"]
    assert np.array_equal(hw.warn_counts(), ow.warn_counts())
    assert np.max(np.abs(ow.get("fxOA"))) > 0                      # the forcings really acted


def test_collision_call_configs4_sparse_100k_fp64():
    """BASELINE configs[4]'s field -- 100 000 floes at 25 % concentration, the broad-phase compaction stress -- in fp64: add_ghosts! +
    timestep_collisions! against the oracle's O(M^2) pair loop: ghosts equal, overlap-pair indices bit-exact, rows and totals 1e-10.  (The
    mixed-precision run of this field is held to this fp64 path: test_mixed_precision_configs4_sparse_100k.)"""
    from subzero_jl_amd import fields
    from subzero_jl_amd import floe as floe_mod
    cfg = fields.make_config(n_floes=100000, seed=12347, concentration=0.25)
    n = cfg["n_floes"]
    # as generated, no two bounding circles of this field touch (that is the broad phase's stress: 5e9 pairs, none to keep).  So that the call
    # also has something to find, every 40th floe is pushed three quarters of the way onto its nearest neighbour -- the rest of the field stays sparse
    cx, cy = cfg["derived"]["cx"].copy(), cfg["derived"]["cy"].copy()
    from scipy.spatial import cKDTree
    movers = np.arange(0, n, 40)
    _, nb = cKDTree(np.stack([cx, cy], 1)).query(np.stack([cx[movers], cy[movers]], 1), k=2)
    off = cfg["vert_off"]
    for m, j in zip(movers, nb[:, 1]):
        dx, dy = 0.75 * (cx[j] - cx[m]), 0.75 * (cy[j] - cy[m])
        cfg["vx"][off[m]:off[m + 1]] += dx; cfg["vy"][off[m]:off[m + 1]] += dy
    cfg["derived"] = floe_mod.derive(cfg["vert_off"], cfg["vx"], cfg["vy"], cfg["height"])
    hw, ow = _pair(cfg); ow.set_threads(_cores())
    hw.add_ghosts(); ow.add_ghosts()
    assert hw.M == ow.M and hw.ghosts() == ow.ghosts()
    hw.timestep_collisions(n, cfg["dt"]); ow.timestep_collisions(n, cfg["dt"])
    res = parity.compare_worlds(hw, ow, rtol=1e-10, fields=["coll_fx", "coll_fy", "coll_trq", "overarea"])
    assert 1000 < res["n_pairs"] < n // 10


def test_reupload_into_the_same_context():
    """A shim uploads before every replaced call: fields of changing size through ONE context (the device chunks of the
    previous upload are carved again, unused ones returned) step exactly like a fresh context does."""
    from subzero_jl_amd import fields
    w = mk()
    for n in (700, 3000, 1200, 3000, 200):
        cfg = fields.make_config(n_floes=n, seed=n)
        fields.build_world(w, cfg)
        w.run(3, 0, cfg["dt"], coupling_dt=1)
        ref = fields.build_world(mk(), cfg)
        ref.run(3, 0, cfg["dt"], coupling_dt=1)
        for f in ("cx", "cy", "u", "v", "xi", "alpha", "coll_fx", "overarea", "fxOA"):
            assert np.array_equal(w.get(f), ref.get(f)), (n, f)
        assert w.stats()["n_pairs"] == ref.stats()["n_pairs"] > n


# ---------------------------------------------------------------- state that must survive uploads / batches that must stop
def test_interactions_survive_a_reupload_of_the_same_size():
    """The shim's per-call pattern: upload, timestep_collisions!, (host work), upload, timestep_floe_properties!.
    calc_stress! (update_floe.jl:392-414) reads the rows the collisions left: they must still be there after the
    second upload (a host-side edit marks the columns dirty, so the second upload really happens)."""
    from subzero_jl_amd import fields
    from subzero_jl_amd.capi import SzError
    cfg = fields.make_config(n_floes=600, seed=31)
    hw, ow = _pair(cfg)
    for w in (hw, ow):
        w.add_ghosts(); w.timestep_collisions(600, cfg["dt"]); w.remove_ghosts(600)
        u = w.get("u"); u[::5] *= 0.5; w.set("u", u)                   # host edit -> dirty -> upload on the next call
        w.timestep_floe_properties(cfg["dt"])
    assert np.abs(ow.get("si11")).max() > 0                            # the stress really comes from contact rows
    parity.compare_worlds(hw, ow, rtol=1e-10, check_pairs=False, check_inter=False)
    # an upload of ANOTHER size drops the rows: the update then refuses to run on nothing ...
    cfg2 = fields.make_config(n_floes=500, seed=32)
    fields.build_world(hw, cfg2)
    hw._new_field = False                                              # (World would declare the new field itself)
    with pytest.raises(SzError, match="interaction rows"):
        hw.timestep_floe_properties(cfg2["dt"])
    # ... until the host says what the interactions are (here: none, fresh floes)
    hw._new_field = True; hw._dirty = True
    hw.timestep_floe_properties(cfg2["dt"])
    assert np.all(hw.get("si11") == 0)


def _tag_scenario(w, kind):
    sq = lambda x0, y0, s=1e4: np.array([[x0, y0], [x0, y0 + s], [x0 + s, y0 + s], [x0 + s, y0], [x0, y0]])
    w.set_consts(E=1e3); w.set_settings()
    if kind == "fuse":          # two floes closing in until their overlap passes floe_floe_max_overlap (collisions.jl:366)
        w.set_domain([1, 1, 1, 1], 0.0, 1e5, 0.0, 1e5)
        rings = [sq(4.0e4, 4.5e4), sq(4.41e4, 4.6e4), sq(8.0e4, 1.0e4), sq(9.6e4, 6.0e4)]      # the last one has a ghost
        u = [3.0, -3.0, 0.0, 0.0]
    else:                       # a floe drifting into an open boundary is tagged remove (collisions.jl:438)
        w.set_domain([0, 0, 0, 0], 0.0, 1e5, 0.0, 1e5)
        rings = [sq(2.0e4, 4.5e4), sq(8.9e4 + 380.0, 2.0e4), sq(5.0e4, 7.0e4)]
        u = [0.0, 20.0, 0.0]
    w.set_grid_fields(10, 10, 0.0, 1e5, 0.0, 1e5, 0.0, 0.0, 0.0, 0.0, 0.0)
    for r in rings:
        w.add_floe(r, 0.5)
    w.set("u", np.array(u))
    return w


def _ab_worlds(build, env_off):
    """two contexts on the same input: the default engine and the one with the switches of env_off in the environment (read at sz_create)"""
    import os
    a = build(mk())
    for k, v in env_off.items():
        os.environ[k] = v
    try:
        b = build(mk())
    finally:
        for k in env_off:
            del os.environ[k]
    return a, b


def _assert_worlds_bit_equal(a, b, fields=None):
    for f in fields or parity.SCALARS:
        assert np.array_equal(a.get(f), b.get(f)), f
    assert np.array_equal(a.rings()[1], b.rings()[1]) and np.array_equal(a.rings()[2], b.rings()[2])
    ia, ib = a.interactions(), b.interactions()
    assert np.array_equal(ia[0], ib[0]) and np.array_equal(ia[1], ib[1])
    assert np.array_equal(a.ids()[2], b.ids()[2]) and a.fuse() == b.fuse()
    pa, pb = a.pairs(), b.pairs()
    assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1])


@pytest.mark.parametrize("scenario", ["dense", "fast-through", "fast-stops", "fuse-stop", "retry-pause", "tagged-before", "walls", "open-stop"])
def test_pipelined_steps_equal_the_three_launch_steps(scenario):
    """Pipelined resident steps (csrc/sz_pipeline.hpp: narrow phase | next geometry, then update | next neighbour search -- two launches per
    timestep) against the three-launch steps (SZ_PIPELINE=0), bit for bit: every column, the rings, floe.interactions, the pair list, the status
    tags and status.fuse_idx.  dense: a plain periodic field; fast-through: fast floes that cross the periodic walls (parents swap with their
    ghosts) and fuse, in batches that run through -- a new tag restarts the enqueued steps so that the next ghosts know it; fast-stops: the same
    with the tag stop (the state handed back has the parents un-swapped again); fuse-stop: the reference's stop on a fuse in the middle of a
    batch; retry-pause: the pause for the largest narrow variant inside a pipelined step; tagged-before: a parent already tagged when the batch
    starts (its first step runs on its own); walls: configs[3]'s kind of field -- four collision walls and the strait's topography, the element
    items of the next step made beside the update; open-stop: a floe drifting through an open boundary (tagged remove, collisions.jl:438)."""
    from subzero_jl_amd import fields
    if scenario in ("dense", "fast-through", "fast-stops", "tagged-before"):
        cfg = fields.make_config(n_floes=1500, seed=77, concentration=0.8)
        def build(w):
            fields.build_world(w, cfg)
            if scenario.startswith("fast"):
                rng = np.random.default_rng(3)
                w.set("u", rng.uniform(-40.0, 40.0, cfg["n_floes"])); w.set("v", rng.uniform(-40.0, 40.0, cfg["n_floes"]))
            if scenario == "tagged-before":
                st = np.ones(cfg["n_floes"], np.int32); st[17] = cases.FUSE
                w.set_status(st)
            return w
        dt = cfg["dt"]
        plan = {"dense": [(25, True), (6, True)], "fast-through": [(9, False), (14, False)], "fast-stops": [(12, True)] * 6, "tagged-before": [(8, True)]}[scenario]
    elif scenario == "walls":
        cfg = fields.make_config(n_floes=900, seed=5, walls=True, topography=True, ocean="strait")
        build = lambda w: fields.build_world(w, cfg); dt = cfg["dt"]; plan = [(20, True), (7, False)]
    elif scenario == "open-stop":
        build = lambda w: _tag_scenario(w, "open"); dt = 10; plan = [(12, True)]
    elif scenario == "fuse-stop":
        build = lambda w: _tag_scenario(w, "fuse"); dt = 10; plan = [(12, True)]
    else:
        build = _retry_scenario; dt = 10; plan = [(8, False), (5, False)]
    a, b = _ab_worlds(build, {"SZ_PIPELINE": "0"})
    t = 0; ran_pipelined = False
    for n, stop in plan:
        coupling = scenario not in ("fuse-stop", "retry-pause", "open-stop")
        da = a.run(n, t, dt, coupling_dt=1 if coupling else 10, coupling_on=coupling, stop_on_tags=stop)
        db = b.run(n, t, dt, coupling_dt=1 if coupling else 10, coupling_on=coupling, stop_on_tags=stop)
        assert da == db and not b.pipelined()
        ran_pipelined |= a.pipelined()
        _assert_worlds_bit_equal(a, b)
        t += da
    assert ran_pipelined, scenario
    if scenario == "fast-stops":
        assert t < 72                                             # batches really ended on tags
    if scenario in ("fuse-stop", "open-stop"):
        assert 2 <= t < 12
    if scenario == "retry-pause":
        assert a.stats()["n_retry"] >= 1


@pytest.mark.parametrize("kind", ["fuse", "open"])
def test_resident_batch_stops_when_a_floe_is_tagged(kind):
    """The reference runs simplify_floes! after EVERY step (simulation.jl:205-214).  A resident batch therefore ends
    after the step that tags a floe, reports how many steps it ran, and leaves the reference's state of that step --
    status.fuse_idx included -- for the host's simplification."""
    hw = _tag_scenario(mk(), kind); ow = _tag_scenario(omk(), kind)
    k = 0
    while k < 12:
        ow.timestep_sim(k, 10, coupling_dt=10, coupling_on=False); k += 1
        if np.any(ow.ids()[2] != cases.ACTIVE):
            break
    assert 2 <= k < 12                                     # the tag comes in the middle of the batch
    done = hw.run(12, 0, 10, coupling_dt=10, coupling_on=False)
    assert done == k
    assert np.array_equal(hw.ids()[2], ow.ids()[2]) and np.any(hw.ids()[2] == (cases.FUSE if kind == "fuse" else cases.REMOVE))
    assert [list(map(int, f)) for f in hw.fuse()] == [list(map(int, f)) for f in ow.fuse()]
    parity.compare_worlds(hw, ow, rtol=1e-9, check_pairs=False)
    st = hw.stats()
    assert st["n_status_fuse"] + st["n_status_remove"] == int(np.sum(ow.ids()[2] != cases.ACTIVE))
    # floes tagged at entry end the next batch after one step (the host has not simplified yet) ...
    assert hw.run(5, k, 10, coupling_dt=10, coupling_on=False) == 1
    # ... unless told to run on regardless
    assert hw.run(3, k + 1, 10, coupling_dt=10, coupling_on=False, stop_on_tags=False) == 3


def _retry_scenario(w):
    """two 8-spike stars drifting into each other: 6, then 8, then 16 crossings -- from the sixth step on the pair outgrows
    the small narrow-phase working set (8 crossings) and is redone by the largest kernel variant"""
    th = np.arange(16) * (2 * np.pi / 16)
    rad = np.where(np.arange(16) % 2 == 0, 1.0e4, 0.55e4)
    star = lambda rot, cx: np.stack([cx + rad * np.cos(-th + rot), 5e4 + rad * np.sin(-th + rot)], 1)
    sq = lambda x0, y0, s=1e4: np.array([[x0, y0], [x0, y0 + s], [x0 + s, y0 + s], [x0 + s, y0], [x0, y0]])
    w.set_consts(E=1e3); w.set_settings()
    w.set_domain([1, 1, 1, 1], 0.0, 1e5, 0.0, 1e5)
    w.set_grid_fields(10, 10, 0.0, 1e5, 0.0, 1e5, 0.0, 0.0, 0.0, 0.0, 0.0)
    w.add_floe(cases.closed(star(0.0, 4.0e4)), 0.5)
    w.add_floe(cases.closed(star(np.pi / 8, 4.5e4)), 0.5)
    w.add_floe(sq(8.0e4, 1.0e4), 0.5); w.add_floe(sq(9.6e4, 6.0e4), 0.5)      # the last one has a ghost
    w.set("u", np.array([0.0, -30.0, 0.0, 0.0]))
    return w


def test_resident_batch_pauses_for_the_largest_narrow_variant():
    """sz_step leaves the largest narrow-phase variant out of its steps until an item needs it: the batch then pauses
    inside that step, the host enqueues the variant, the rest of the step and the remaining steps.  The trajectory must
    be the oracle's, and bit for bit the one of a run that enqueues the variant in every step."""
    import os
    from oracle import orc
    ow = _retry_scenario(omk())
    first = None
    for k in range(10):
        o, x, y = ow.rings()
        if first is None and len(orc.intersection_points(np.stack([x[o[0]:o[1]], y[o[0]:o[1]]], 1), np.stack([x[o[1]:o[2]], y[o[1]:o[2]]], 1))) > 8:
            first = k
        ow.timestep_sim(k, 10, coupling_dt=10, coupling_on=False)
    assert first is not None and 2 <= first < 8                 # the pause comes in the middle of the first batch
    runs = []
    for lean in ("1", "0"):
        os.environ["SZ_LEAN_NARROW"] = lean
        try:
            hw = _retry_scenario(mk())
        finally:
            del os.environ["SZ_LEAN_NARROW"]
        assert hw.run(8, 0, 10, coupling_dt=10, coupling_on=False, stop_on_tags=False) == 8
        assert hw.stats()["n_retry"] >= 1
        assert hw.run(2, 8, 10, coupling_dt=10, coupling_on=False, stop_on_tags=False) == 2      # the variant stays in
        runs.append(hw)
    parity.compare_worlds(runs[0], ow, rtol=1e-9, check_pairs=False)
    for f in parity.SCALARS:
        assert np.array_equal(runs[0].get(f), runs[1].get(f)), f
    assert np.array_equal(runs[0].rings()[1], runs[1].rings()[1])
    a, b = runs[0].interactions(), runs[1].interactions()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("kind", ["periodic", "walls", "fast", "mixed"])
def test_collision_records_stay_current(kind):
    """The collision records (State::crec: one 128-byte line per floe with what the neighbour search and the narrow phase's staging read of it) are
    a CACHE of the columns, kept current inside a resident batch by whoever places a floe (integrator, inline ghost maker; seeded at the start
    of the batch).  After batches of several kinds every record of every parent must equal the columns bit for bit -- also where nothing in
    the results would show a stale one yet: a periodic field, walls + topography (no ghosts at all), fast floes that cross the periodic walls
    (parents swap with their ghosts), mixed precision (body-frame rings)."""
    from subzero_jl_amd import fields
    if kind == "walls":
        cfg = fields.make_config(n_floes=900, seed=5, walls=True, topography=True, ocean="strait")
    else:
        cfg = fields.make_config(n_floes=1200, seed=21, concentration=0.8)
    hw = fields.build_world(mk(), cfg)
    if kind == "fast":
        rng = np.random.default_rng(3)
        hw.set("u", rng.uniform(-40.0, 40.0, cfg["n_floes"])); hw.set("v", rng.uniform(-40.0, 40.0, cfg["n_floes"]))
    if kind == "mixed":
        hw.set_precision("mixed")
    t = 0
    for k in (1, 7, 3, 1, 12):
        assert hw.run(k, t, cfg["dt"], coupling_dt=1, stop_on_tags=False) == k
        t += k
        assert hw.crec_mismatches() == 0, (kind, t)
    if kind == "fast":
        # (this scenario is also the one that found a latent host bug: batches that run on past a fuse -- stop_on_tags=False -- replayed fuse lists
        #  that still named the ghost numbers of earlier batches, and wrote past the end of the replay's arrays.  The lists of a batch describe
        #  the step that ended it: every entry names a floe of that step.)
        st = hw.stats()
        assert st["n_ghosts"] > 0 and st["n_status_fuse"] > 50
        lists = hw.fuse()
        assert sum(len(l) for l in lists) > 0 and all(0 <= int(v) < cfg["n_floes"] + st["n_ghosts"] for l in lists for v in l)


def test_float32_host_boundary():
    """The _f32 instantiation of the boundary (sz_upload_floes_f32, sz_download_floes_f32, sz_set_fields_f32, sz_download_interactions_f32:
    SURVEY section 8b; Floe{FT} is generic, floe.jl:24).  documentation.md:25 supports Float64 only -- there are no Float32 answers -- so the
    contract is: columns are widened on the way in, the engine computes as for a Float64 host, results are rounded on the way out.  A
    Float32 host must therefore get exactly the Float64 host's results for the same (float-representable) inputs, rounded once."""
    import subzero_jl_amd
    from subzero_jl_amd import fields, capi
    r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
    cfg = fields.make_config(n_floes=600, seed=31, concentration=0.8, ocean="converge_diverge")
    for k in ("vx", "vy", "sx", "sy", "u", "v", "xi", "uo", "vo", "hf", "ua", "va"):
        cfg[k] = r32(cfg[k])
    cfg["derived"] = {k: (r32(v) if np.asarray(v).dtype == np.float64 else v) for k, v in cfg["derived"].items()}
    w32 = fields.build_world(subzero_jl_amd.World(0, np.float32), cfg)
    w64 = fields.build_world(subzero_jl_amd.World(0), cfg)
    for w in (w32, w64):
        assert w.run(6, 0, cfg["dt"], coupling_dt=1, stop_on_tags=False) == 6
    for f in parity.SCALARS + ["fxOA", "fyOA", "trqOA", "coll_fx", "coll_fy", "coll_trq", "overarea"]:
        a, b = w32.get(f), w64.get(f)
        assert np.array_equal(a, r32(b)), f
    assert not np.array_equal(w64.get("cx"), r32(w64.get("cx")))          # (the Float64 host's columns really carry more digits)
    o32, x32, y32 = w32.rings(); o64, x64, y64 = w64.rings()
    assert np.array_equal(o32, o64) and np.array_equal(x32, r32(x64)) and np.array_equal(y32, r32(y64))
    i32, i64 = w32.interactions(), w64.interactions()
    assert np.array_equal(i32[0], i64[0]) and i64[0][-1] > 100 and np.array_equal(i32[1], r32(i64[1]))


def test_one_step_batches_match_the_oracle():
    """timestep_sim! called step by step (batches of one resident step each, nothing in between): what a batch leaves
    behind -- cell lists, ghost bookkeeping of its last integrator -- must not leak into the next one"""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=400, seed=7)
    hw, ow = _pair(cfg)
    for t in range(4):
        hw.timestep_sim(t, cfg["dt"], coupling_dt=1); ow.timestep_sim(t, cfg["dt"], coupling_dt=1)
    parity.compare_worlds(hw, ow, rtol=1e-9)
    hw.run(3, 4, cfg["dt"], coupling_dt=1, stop_on_tags=False)
    for t in range(4, 7):
        ow.timestep_sim(t, cfg["dt"], coupling_dt=1)
    hw.timestep_sim(7, cfg["dt"], coupling_dt=1); ow.timestep_sim(7, cfg["dt"], coupling_dt=1)
    parity.compare_worlds(hw, ow, rtol=1e-9)


def test_random_call_sequences_match_the_oracle():
    """tools/fuzz_sequences.py, a few seeds of it: random sequences of resident batches, timestep_sim! step by step, the process-mode
    sequence, host edits and downloads on small periodic fields with fast floes (parents cross the walls), against the oracle at
    every checkpoint -- nothing a call leaves behind may leak into the next (300 seeds x 15 calls were run when this was written:
    all agree to the round-off of 30-step runs)"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_sequences", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_sequences.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    assert mod.run(nseeds=10, nops=10, rtol=1e-8, verbose=False) == 0
    assert mod.run(nseeds=8, nops=10, rtol=1e-8, verbose=False, walls=True) == 0          # ... and between collision walls with topography


def test_field_reupload_keeps_the_temperatures():
    """sz_set_fields with an unchanged lattice shape keeps the ocean / atmosphere temperatures of sz_set_temps (the
    heat-flux factor of calc_two_way_coupling!, coupling.jl:1676, depends on them)."""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=300, seed=41, ocean="shear")
    L, Nx, Ny = cfg["L"], cfg["Nx"], cfg["Ny"]
    res = []
    for again in (False, True):
        w = fields.build_world(mk(), cfg)
        w.set_two_way(True, dt=cfg["dt"]); w.set_temps(0.7, -12.0)
        w.run(2, 0, cfg["dt"], coupling_dt=1)
        w.set_grid_fields(Nx, Ny, 0.0, L, 0.0, L, 0.5 * cfg["uo"], cfg["vo"] + 0.05, cfg["hf"], cfg["ua"], cfg["va"])
        if again:
            w.set_temps(0.7, -12.0)
        w.run(2, 2, cfg["dt"], coupling_dt=1)
        res.append((w.ocean_stress(), w.get("height"), w.get("u")))
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert np.any(res[0][0][3] != 0)                      # a heat-flux factor that is not the zero-temperature one


def test_ghost_candidate_list_equals_the_two_launch_pass():
    """The resident steps make the periodic ghosts from a candidate list kept by the integrator (one launch); the
    process-mode pass (flag + scan over all parents, fill) is pinned by the reference's add_ghosts! cases.  Both must
    give the same run bit for bit -- fast floes, so that parents cross the walls and swap with their ghosts
    (collisions.jl:942-950), and a host edit in the middle (the list is seeded again)."""
    import os
    from subzero_jl_amd import fields
    from subzero_jl_amd import floe as floe_mod
    cfg = fields.make_config(n_floes=500, seed=77)
    cfg["u"] = np.abs(cfg["u"]) * 60.0 + 2.0; cfg["v"] = cfg["v"] * 60.0        # eastward, 2 to 8 m/s: up to 160 m per step
    # the lattice keeps centroids 8 km off the walls: shift the field so that the last column's centroids straddle the
    # east wall (some parents start outside the domain and swap in step 0, others cross during the run)
    cfg["vx"] = cfg["vx"] + 9800.0; cfg["vy"] = cfg["vy"] + 9900.0
    cfg["derived"] = floe_mod.derive(cfg["vert_off"], cfg["vx"], cfg["vy"], cfg["height"])
    runs = []
    for env in ("1", "0"):
        os.environ["SZ_GHOST_LIST"] = env
        try:
            w = fields.build_world(mk(), cfg)
        finally:
            del os.environ["SZ_GHOST_LIST"]
        w.run(40, 0, cfg["dt"], coupling_dt=1, stop_on_tags=False)
        u = w.get("u"); u[::9] *= -1.0; w.set("u", u)
        w.run(35, 40, cfg["dt"], coupling_dt=1, stop_on_tags=False)
        w.run(5, 75, cfg["dt"], coupling_dt=1, stop_on_tags=False)
        runs.append({f: w.get(f) for f in ("cx", "cy", "u", "v", "xi", "alpha", "coll_fx", "coll_fy", "overarea", "sa11")} | {"rings": w.rings()[1], "ng": w.stats()["n_ghosts"]})
    a, b = runs
    assert a["ng"] == b["ng"] > 20
    for f in a:
        assert np.array_equal(a[f], b[f]), f
    L = cfg["L"]
    wrapped = np.abs(a["cx"] - cfg["derived"]["cx"]) > 0.5 * L
    assert wrapped.sum() >= 3                            # parents went through the east wall and came back in the west
    assert np.any(wrapped & (cfg["derived"]["cx"] <= L)) # ... also some that started inside


# ---------------------------------------------------------------- further reference-held vectors, through the C-ABI
def test_which_vertices_match_points(golden):
    """test_floe_utils.jl:74-137, by the narrow phase's own device routine"""
    w = mk()
    for case in golden["floe_utils"]["which_vertices_match_points"]:
        assert w.which_vertices_match_points(case["points"], case["region"]) == case["expected"], case["name"]


def test_translate_rotate(golden):
    """test_floe_utils.jl:52-63, 173-192 through _move_floe! of timestep_floe_properties!"""
    cases.check_translate_rotate(mk, golden["floe_utils"])


def test_boundary_rectangles_and_update(golden):
    """boundaries.jl (test):5-83 and :103-127 (_update_boundary!: only MovingBoundary walls move)"""
    for key in ("directions", "boundaries"):
        B = golden["boundaries"][key]
        x0, xf, y0, yf = B["extent"]
        w = mk(); w.set_domain([0, 0, 0, 0], x0, xf, y0, yf)
        cases.check_boundary_polys(w.boundary_polys(), w.boundary_vals(), B)
    U = golden["boundaries"]["update"]
    cases.check_update_boundaries(cases.run_update_boundaries(mk, U), U)


@pytest.mark.parametrize("k", range(5))
def test_conservation(golden, k):
    """test_conservation.jl:58-203 through resident batches: < 1 % change of kinetic energy, linear and angular momentum over
    the reference's 5000 steps (2.1 % for the three many-sided non-convex floes of floe_shapes.jld2, rings of 50 / 146 / 203 points;
    energy only for the 35-point floe next to a wall and a topography element).  Read closely, NONE of the reference's five runs
    brings its floes into contact (10 km apart at 0.25 m/s, dt = 1 s; the complex shapes pass one another, the last floe moves
    AWAY from the topography): the criterion pins the free-flight AB2 update, update_floe.jl:502-545.  So every case is also
    run in a variant in which the floes DO collide for thousands of steps -- dt = 10 s for the literal blocks; the complex
    shapes sent into one another (rings far above 32 points: the 64-lane narrow variant, multi-region contacts), and the
    non-convex floe sent into the topography between collision walls (the boundary clip) -- and there the HIP path is held to
    the oracle on the same four quantities."""
    C = golden["conservation"]; case = C["cases"][k]

    def stepper(w, n, dt):
        t = 0
        while t < n:
            done = w.run(min(500, n - t), t, dt, coupling_dt=10, coupling_on=False)
            assert done > 0
            t += done
    change = cases.run_conservation(mk, C, case, stepper)
    assert cases.conservation_ok(C, case, change), (case["name"], change)
    # colliding variant
    Cv, cv = dict(C), dict(case)
    if case["name"] == "complex_shapes":          # floe 3 east into floe 4, floe 5 north into floe 4
        cv.update(u=[0.5, 0.0, 0.0], v=[0.0, 0.0, 1.0])
    elif case["name"] == "wall_and_topography":   # north-east into the topography's long side, then back onto the walls
        cv.update(u=[0.09], v=[0.09]); Cv["boundaries"] = "collision"
    res = []
    for make, step in ((mk, lambda w, n, dt: stepper(w, n, 10)),
                       (omk, lambda w, n, dt: [w.timestep_sim(t, 10, coupling_dt=10, coupling_on=False) for t in range(n)])):
        res.append(cases.run_conservation(make, Cv, cv, step))
    assert np.any(np.abs(res[1]) > 1.0)                      # contacts really happened (energy is not conserved by them)
    assert np.allclose(res[0], res[1], rtol=1e-6, atol=1e-9), (case["name"], res)


def test_c_example_links_and_runs(tmp_path):
    """examples/minimal.c: the C-ABI from plain C, linked against libsubzero_hip.so and RUN on the GPU (the CPU suite only
    compiles it)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "subzero.jl_amd")
    exe = str(tmp_path / "minimal")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "minimal.c"),
                           "-L" + libdir, "-lsubzero_hip", "-L/opt/rocm/lib", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
                           "-Wl,--allow-shlib-undefined", "-lm", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    lines = out.stdout.splitlines()
    assert lines[0].startswith("pairs 1, contact rows 2"), lines
    u = [float(x) for x in lines[1].split(":")[1].split("(")[0].split()]
    assert u[0] < 0.1 and u[1] > -0.1                      # the two floes have pushed each other apart
    assert "0 remove, 0 fuse" in lines[2]
    assert abs(float(lines[3].split(":")[1].split()[0]) - 2e8) < 1e-3 * 2e8      # the grid output adds up to the ice area


def _size_spectrum(w, n_small, big_r):
    """one large floe with n_small small ones around its rim, each overlapping it a little (a Voronoi field with a size
    spectrum in miniature): the large floe has n_small neighbours, every broad-phase cell holds dozens of floes"""
    w.set_consts(E=1e5); w.set_settings()
    w.set_domain([0, 0, 0, 0], -4e5, 4e5, -4e5, 4e5)
    w.set_grid_fields(8, 8, -4e5, 4e5, -4e5, 4e5, 0.0, 0.0, 0.0, 0.0, 0.0)
    th = -np.arange(24) * (2 * np.pi / 24)
    w.add_floe(cases.closed(np.stack([big_r * np.cos(th), big_r * np.sin(th)], 1)), 0.5)
    small_r = 0.9 * np.pi * big_r / n_small
    t6 = -np.arange(6) * (2 * np.pi / 6)
    for k in range(n_small):
        a = 2 * np.pi * k / n_small
        c = (big_r + 0.6 * small_r) * np.array([np.cos(a), np.sin(a)])
        w.add_floe(cases.closed(c + small_r * np.stack([np.cos(t6 + a), np.sin(t6 + a)], 1)), 0.3)
    u = np.zeros(n_small + 1); v = np.zeros(n_small + 1)
    ang = 2 * np.pi * np.arange(n_small) / n_small
    u[1:] = -0.2 * np.cos(ang); v[1:] = -0.2 * np.sin(ang)            # the small floes press inward
    w.set("u", u); w.set("v", v)
    return w


def test_size_spectrum_field_and_its_capacity_limit():
    """A large floe among small ones (the reference's Voronoi fields have a size spectrum).  The library counts the
    bounding-circle neighbours of the uploaded field and sizes its neighbour lists from that: 24 per floe and direction for
    like-sized floes, 64 (and 128 interaction rows per floe, the chunked candidate pool) where a floe has more -- the contact
    rows match the oracle either way, crowded cells (bucket + overflow chain) included.  Beyond 64: the third capacity, 256
    (a quarter of the lane groups per workgroup: the capacity that keeps such a field running, not a fast path); beyond that the
    step fails LOUDLY (SZ_E_CAPACITY, neighbours bit) instead of dropping contacts."""
    from subzero_jl_amd.capi import SzError
    hw = _size_spectrum(mk(), 20, 3.0e4); ow = _size_spectrum(omk(), 20, 3.0e4)
    for w in (hw, ow):
        w.timestep_collisions(21, 10)
    assert parity.compare_pairs(hw, ow) >= 20
    parity.compare_interactions(hw, ow, 1e-10)
    parity.compare_worlds(hw, ow, rtol=1e-10, fields=["coll_fx", "coll_fy", "coll_trq", "overarea"])
    assert len(ow.inter(0)) >= 20                          # the large floe really carries all those contacts
    hw.run(5, 0, 10, coupling_on=False); [ow.timestep_sim(t, 10, coupling_dt=10, coupling_on=False) for t in range(5)]
    parity.compare_worlds(hw, ow, rtol=1e-9)
    # 40 small floes around the large one: more than the default lists hold -- the upload picks the larger capacities
    hw, ow = _size_spectrum(mk(), 40, 6.0e4), _size_spectrum(omk(), 40, 6.0e4)
    for w in (hw, ow):
        w.timestep_collisions(41, 10)
    assert parity.compare_pairs(hw, ow) >= 40
    parity.compare_interactions(hw, ow, 1e-10)
    assert len(ow.inter(0)) >= 40
    hw.run(5, 0, 10, coupling_on=False); [ow.timestep_sim(t, 10, coupling_dt=10, coupling_on=False) for t in range(5)]
    parity.compare_worlds(hw, ow, rtol=1e-9)
    # 90 small floes: beyond 64 neighbours -- the third capacity (256 per floe and direction)
    hw, ow = _size_spectrum(mk(), 90, 1.4e5), _size_spectrum(omk(), 90, 1.4e5)
    for w in (hw, ow):
        w.timestep_collisions(91, 10)
    assert parity.compare_pairs(hw, ow) >= 90
    parity.compare_interactions(hw, ow, 1e-10)
    assert len(ow.inter(0)) >= 90
    hw.run(5, 0, 10, coupling_on=False); [ow.timestep_sim(t, 10, coupling_dt=10, coupling_on=False) for t in range(5)]
    parity.compare_worlds(hw, ow, rtol=1e-9)


@pytest.mark.parametrize("n_small,start,rowcap", [(40, "24", None), (90, "24", None), (20, "24", "8"), (40, "64", "16")])
def test_lists_grow_when_a_step_outgrows_them(monkeypatch, n_small, start, rowcap):
    """The reference's lists grow as needed (collisions.jl:290-296: interactions by vcat; the Dict of the pair loop).  The engine's
    neighbour lists, pair items and interaction rows are carved per upload from a count of the field -- and when a call or a step
    outgrows them all the same (here: the upload is TOLD to start too small, SZ_MAXNB / SZ_ROWCAP), the library carves larger ones
    and runs the call / the step again: process-mode timestep_collisions! and resident batches (the batch pauses in the step that
    overflowed, before anything of the floes' state has changed) both match the oracle, floe.overarea is not added twice."""
    monkeypatch.setenv("SZ_MAXNB", start)
    if rowcap:
        monkeypatch.setenv("SZ_ROWCAP", rowcap)
    big_r = {20: 3.0e4, 40: 6.0e4, 90: 1.4e5}[n_small]
    hw, ow = _size_spectrum(mk(), n_small, big_r), _size_spectrum(omk(), n_small, big_r)
    for w in (hw, ow):
        w.timestep_collisions(n_small + 1, 10)
    assert parity.compare_pairs(hw, ow) >= n_small
    parity.compare_interactions(hw, ow, 1e-10)
    parity.compare_worlds(hw, ow, rtol=1e-10, fields=["coll_fx", "coll_fy", "coll_trq", "overarea"])
    # a second call on the grown lists, then a resident batch on a FRESH context that has to grow inside its first step
    for w in (hw, ow):
        w.timestep_collisions(n_small + 1, 10)
    parity.compare_worlds(hw, ow, rtol=1e-10, fields=["coll_fx", "coll_fy", "coll_trq", "overarea"])
    hw, ow = _size_spectrum(mk(), n_small, big_r), _size_spectrum(omk(), n_small, big_r)
    assert hw.run(5, 0, 10, coupling_on=False) == 5
    [ow.timestep_sim(t, 10, coupling_dt=10, coupling_on=False) for t in range(5)]
    parity.compare_worlds(hw, ow, rtol=1e-9)
    assert np.array_equal(hw.warn_counts(), ow.warn_counts())


# ---------------------------------------------------------------- mixed precision (BASELINE configs[4])
def _mixed_vs_f64(cfg, steps):
    from subzero_jl_amd import fields
    h64 = fields.build_world(mk(), cfg); h32 = fields.build_world(mk(), cfg)
    h32.set_precision("mixed")
    h64.run(steps, 0, cfg["dt"], coupling_dt=1, stop_on_tags=False); h32.run(steps, 0, cfg["dt"], coupling_dt=1, stop_on_tags=False)
    return h64, h32


def _mixed_acceptance(h64, h32, cfg):
    """The reference has no Float32 answers (documentation.md:25: only Float64 is tested and supported), so mixed precision
    -- fp32 broad-phase records with fp64 confirmation, fp32 body-frame rings + fp64 pose with the narrow-phase predicates
    in fp64 on the reconstructed coordinates, fp32 per-point forcings -- is held to the engine's own fp64 path:
      * the pair list (bounding-circle test, Dict rule) is the fp64 one BIT FOR BIT while the positions agree;
      * discrete decisions of the narrow phase (rows per pair, partner of every row) agree on >= 99.9 % of the pairs;
      * on the agreeing pairs forces agree to 1e-5 relative above a geometric floor: a ring vertex reconstructed from fp32
        body offsets is off by <= 1 mm (half an fp32 spacing at 16 km), which moves an overlap area by <= contact length x 1 mm;
      * state: velocities to 1e-5 and angular velocities to 2e-4 of the column's scale (contact forces that differ by 1e-5
        move them that much), positions to a centimetre."""
    o64, r64 = h64.interactions(); o32, r32 = h32.interactions()
    n = len(o64) - 1
    c64, c32 = np.diff(o64), np.diff(o32)
    same = c64 == c32
    for i in np.nonzero(same)[0]:
        if c64[i] and not np.array_equal(r64[o64[i]:o64[i + 1], 0], r32[o32[i]:o32[i + 1], 0]):
            same[i] = False
    assert same.mean() >= 0.999, same.mean()
    idx = np.nonzero(same & (c64 > 0))[0]
    a = np.concatenate([r32[o32[i]:o32[i + 1]] for i in idx]) if len(idx) else np.zeros((0, 7))
    b = np.concatenate([r64[o64[i]:o64[i + 1]] for i in idx]) if len(idx) else np.zeros((0, 7))
    if len(b):
        per_area = np.max(np.hypot(b[:, 1], b[:, 2]) / b[:, 6])
        floor = per_area * (4.0 * np.sqrt(cfg["derived"]["area"].max())) * 2e-3          # contact length <= the floe's size, 2 x 1 mm
        for col in (1, 2):
            parity.assert_elementwise(f"force column {col}", a[:, col], b[:, col], 1e-5, floor)
    for k, tol in (("u", 1e-5), ("v", 1e-5), ("xi", 2e-4)):       # (xi: torques are lever arm x force, the force points move by millimetres)
        assert np.max(np.abs(h32.get(k) - h64.get(k))) <= tol * np.max(np.abs(h64.get(k))), k
    assert np.max(np.abs(h32.get("cx") - h64.get("cx"))) < 1e-2 and np.max(np.abs(h32.get("cy") - h64.get("cy"))) < 1e-2
    return len(b)


def test_mixed_precision_dense_field():
    """contacts everywhere: one step from identical state (pair list bit-exact), then the acceptance criterion after 5"""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=2000, seed=61)
    h64, h32 = _mixed_vs_f64(cfg, 1)
    assert parity.compare_pairs(h32, h64) > 2000                    # bit-exact: same positions, fp64-confirmed circle test
    assert _mixed_acceptance(h64, h32, cfg) > 1000
    h64, h32 = _mixed_vs_f64(cfg, 5)
    _mixed_acceptance(h64, h32, cfg)
    # it really ran on body-frame rings: the rings that come back are rebuilt from the fp32 offsets (not bit-equal), to a millimetre
    x64, x32 = h64.rings()[1], h32.rings()[1]
    assert not np.array_equal(x64, x32) and np.max(np.abs(x64 - x32)) < 1.1e-2
    # process-mode calls and further batches go on from the rebuilt rings
    for w in (h64, h32):
        w.add_ghosts(); w.timestep_collisions(2000, cfg["dt"]); w.remove_ghosts(2000); w.timestep_floe_properties(cfg["dt"])
        w.run(3, 6, cfg["dt"], coupling_dt=1, stop_on_tags=False)
    _mixed_acceptance(h64, h32, cfg)


def test_mixed_precision_configs4_sparse_100k():
    """BASELINE configs[4]: 100 000 floes at 25 % concentration (broad-phase compaction stress), mixed precision against the
    fp64 path: pair list bit-exact after the first step, the acceptance criterion after 10"""
    from subzero_jl_amd import fields
    cfg = fields.make_config(n_floes=100000, seed=12347, concentration=0.25)
    h64, h32 = _mixed_vs_f64(cfg, 1)
    a, b = h32.pairs(), h64.pairs()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    h64.run(9, 1, cfg["dt"], coupling_dt=1, stop_on_tags=False); h32.run(9, 1, cfg["dt"], coupling_dt=1, stop_on_tags=False)
    _mixed_acceptance(h64, h32, cfg)
    assert h32.stats()["n_trace_fail"] == 0
