"""The CPU oracle against the reference's own known-answer tests (tests/golden/*.json).

This pins the oracle: every literal-input test of the reference's
test_collisions.jl, the moment-of-inertia values of test_floe_utils.jl and the OA-forcing
values of test_coupling.jl and the stress / strain values of test_update_floe.jl must be reproduced before the oracle is trusted as the checker
for the HIP path.
"""
import numpy as np
import pytest

import cases
from oracle import orc


def mk():
    return orc.World()


def test_clip_basic_square_overlap():
    a = np.array([[0, 0], [0, 2], [2, 2], [2, 0], [0, 0]], float)
    b = a + 1.0
    regs = orc.clip(a, b)
    assert len(regs) == 1
    x, y = regs[0][:, 0], regs[0][:, 1]
    area = 0.5 * abs(np.sum(x[:-1] * y[1:] - x[1:] * y[:-1]))
    assert area == 1.0
    assert len(orc.intersection_points(a, b)) == 2
    # orientation independence
    regs2 = orc.clip(a[::-1].copy(), b)
    assert len(regs2) == 1
    # disjoint / contained
    assert orc.clip(a, a + 10.0) == []
    inner = np.array([[0.5, 0.5], [0.5, 1.5], [1.5, 1.5], [1.5, 0.5], [0.5, 0.5]])
    r = orc.clip(a, inner)
    assert len(r) == 1 and np.array_equal(r[0], inner)
    r = orc.clip(inner, a)
    assert len(r) == 1 and np.array_equal(r[0], inner)


def test_moment_of_inertia(golden):
    for c in golden["floe_utils"]["moment"]:
        w = mk()
        w.add_floe(np.array(c["coords"], float), c["height"])
        assert abs(w.get("moment")[0] - c["expected"]) < c["atol"]


@pytest.mark.parametrize("k", range(5))
def test_floe_floe(golden, k):
    G = golden["collisions"]
    case = G["floe_floe"][k]
    cases.check_floe_floe(cases.run_floe_floe(mk, G, case), case)


@pytest.mark.parametrize("k", range(8))
def test_floe_boundary(golden, k):
    G = golden["collisions"]
    case = G["boundary"]["cases"][k]
    cases.check_boundary(cases.run_boundary(mk, G, case), case)


@pytest.mark.parametrize("k", range(4))
def test_add_ghosts(golden, k):
    G = golden["collisions"]
    case = G["add_ghosts"]["cases"][k]
    cases.check_add_ghosts(cases.run_add_ghosts(mk, G, case), G, case)


def test_ghost_collisions(golden):
    G = golden["collisions"]
    cases.check_ghost_collisions(cases.ghost_collision_scenarios(mk, G), exact=True)


@pytest.mark.parametrize("k", range(6))
def test_forcings(golden, k):
    F = golden["forcings"]
    case = F["cases"][k]
    cases.check_forcing(cases.run_forcing(mk, F, case), case)


def test_stress_strain(golden):
    U = golden["update_floe"]
    cases.check_stress_strain(cases.run_stress_strain(mk, U), U)
