"""CPU-side checks of the product's host logic and of the C-ABI library itself (no compute calls:
there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

import subzero_jl_amd
from subzero_jl_amd import capi, fields, floe
from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = capi.load()
    hdr = open(os.path.join(ROOT, "include", "subzero_hip.h")).read()
    declared = set(re.findall(r"\b(sz_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/subzero_hip.h but not exported"
    assert set(capi.EXPORTS) <= declared
    assert b"gfx950" in L.sz_version()


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(subzero_jl_amd.SzError):
        subzero_jl_amd.World(0)


def test_product_does_not_touch_the_oracle():
    """the shipped path must never import, link or execute anything under oracle/"""
    pkg = os.path.join(ROOT, "subzero.jl_amd")
    bad = re.compile(r"(from\s+oracle|import\s+oracle|liborc|orc\.h|orc_[a-z_]+\s*\(|oracle/)")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not bad.search(src), f


def test_derived_columns_match_oracle_constructor():
    cfg = fields.make_config(n_floes=150, seed=11)
    w = fields.build_world(orc.World(), cfg)
    for k in ("cx", "cy", "area", "mass", "moment", "rmax"):
        assert np.array_equal(w.get(k), cfg["derived"][k]), k


def test_valid_ring_and_boundary_rects():
    r = floe.valid_ring([[0, 0], [0, 1], [0, 1], [1, 1], [1, 0]])
    assert len(r) == 5 and np.array_equal(r[0], r[-1])
    rects, vals = floe.boundary_rects(-1e5, 1e5, -1e5, 1e5)
    assert list(vals) == [1e5, -1e5, 1e5, -1e5]
    assert list(rects[0]) == [-2e5, 2e5, 1e5, 2e5] and list(rects[3]) == [-2e5, -1e5, -2e5, 2e5]


def test_oracle_thread_count_does_not_change_results():
    cfg = fields.make_config(n_floes=300, seed=5)
    res = []
    for nt in (1, 4):
        w = fields.build_world(orc.World(), cfg)
        w.set_threads(nt)
        for t in range(3):
            w.timestep_sim(t, cfg["dt"], coupling_dt=1)
        res.append((w.get("cx"), w.get("u"), w.get("coll_fx"), w.interactions()[1]))
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


def test_oracle_momentum_bookkeeping():
    """internal contact forces cancel over the parents (Newton's third law through mirror + ghost fold)"""
    cfg = fields.make_config(n_floes=400, seed=9)
    w = fields.build_world(orc.World(), cfg)
    w.add_ghosts()
    w.timestep_collisions(400, cfg["dt"])
    fx, fy = w.get("coll_fx")[:400], w.get("coll_fy")[:400]
    scale = np.abs(fx).max()
    assert abs(fx.sum()) < 1e-9 * scale * 400 and abs(fy.sum()) < 1e-9 * scale * 400
    assert np.all(w.get("coll_fx")[400:] == 0)


def test_header_is_plain_c_and_the_c_example_compiles():
    """include/subzero_hip.h must be consumable by a C compiler (the boundary is a C-ABI): the C example is
    compiled (not linked, not run: no GPU here) with gcc -std=c11 -Wall -Werror."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"), "-fsyntax-only",
                        os.path.join(root, "examples", "minimal.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
