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


def _julia_structs():
    """the C-ABI mirrors of julia/SubzeroHIP.jl: {struct name: [(field, julia type)]}"""
    src = open(os.path.join(ROOT, "julia", "SubzeroHIP.jl")).read()
    out = {}
    for m in re.finditer(r"^(?:mutable )?struct (Sz\w+)\n(.*?)^end", src, re.S | re.M):
        out[m.group(1)] = re.findall(r"^\s+(\w+)::([\w{}]+)\s*$", m.group(2), re.M)
    return out


def test_julia_struct_mirrors_match_the_header(tmp_path):
    """julia/SubzeroHIP.jl cannot be run here (no Julia in the image).  What CAN be checked without Julia: its three
    struct mirrors have the same fields, in the same order, at the same offsets and of the same sizes as the C structs a
    C compiler lays out from include/subzero_hip.h (Julia lays out isbits / pointer fields like C does)."""
    import subprocess
    js = _julia_structs()
    # (the transport of tiled runs is four pointers: user + three function pointers, checked on its own below)
    tr = js.pop("SzHostTransport")
    assert [f for f, _ in tr] == ["user", "allgather", "sendrecv", "allreduce_sum_f64"] and all(t == "Ptr{Cvoid}" for _, t in tr)
    assert ctypes.sizeof(capi.SzHostTransport) == 32 and [f for f, _ in capi.SzHostTransport._fields_] == [f for f, _ in tr]
    cnames = {"SzParams": "sz_params", "SzFloeColumns": "sz_floe_columns", "SzFloeColumnsF32": "sz_floe_columns_f32", "SzStats": "sz_stats"}
    assert set(js) == set(cnames)
    size_of = {"Float64": 8, "Int64": 8, "Int32": 4}
    hdr = open(os.path.join(ROOT, "include", "subzero_hip.h")).read()
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "subzero_hip.h"', 'int main(void) {']
    expect = []
    for jn, cn in cnames.items():
        # every member the header declares must be mirrored: count the declarators of the C struct
        body = {m.group(2): m.group(1) for m in re.finditer(r"typedef struct \{([^{}]*)\}\s*(\w+)\s*;", hdr)}[cn]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        cfields = [f.strip().lstrip("*") for decl in body.split(";") if decl.strip()
                   for f in re.sub(r"^\s*(?:const\s+)?\w+\s+", "", decl.strip()).split(",")]
        assert [f for f, _ in js[jn]] == cfields, (jn, cfields)
        off = 0
        for f, t in js[jn]:
            sz = 8 if t.startswith("Ptr{") else size_of[t]
            off = (off + sz - 1) // sz * sz
            expect.append(f"{cn}.{f} {off} {sz}")
            prog.append(f'  printf("{cn}.{f} %zu %zu\\n", offsetof({cn}, {f}), sizeof((({cn}*)0)->{f}));')
            off += sz
        expect.append(f"{cn} {(off + 7) // 8 * 8}")
        prog.append(f'  printf("{cn} %zu\\n", sizeof({cn}));')
    prog += ["  return 0;", "}"]
    src = tmp_path / "layout.c"; exe = tmp_path / "layout"
    src.write_text("\n".join(prog))
    subprocess.check_call(["gcc", "-std=c11", "-I" + os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = subprocess.check_output([str(exe)], text=True).split("\n")
    assert [g for g in got if g] == expect
    # ... and the ctypes mirrors of the Python binding agree with the same C layout
    assert ctypes.sizeof(capi.SzParams) == int(expect[[e.split()[0] for e in expect].index("sz_params")].split()[1])
    assert ctypes.sizeof(capi.SzFloeColumns) == 39 * 8 and ctypes.sizeof(capi.SzFloeColumnsF32) == 39 * 8 and ctypes.sizeof(capi.SzStats) == 29 * 8


def test_julia_shim_binds_existing_symbols():
    """every @ccall in julia/SubzeroHIP.jl names a function the header declares, with the declared number of arguments"""
    src = open(os.path.join(ROOT, "julia", "SubzeroHIP.jl")).read()
    hdr = open(os.path.join(ROOT, "include", "subzero_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    decl = {m.group(1): m.group(2) for m in re.finditer(r"\b(sz_\w+)\s*\(([^;{]*?)\)\s*;", hdr)}
    calls = []
    for m in re.finditer(r"@ccall lib\.(sz_\w+)\(", src):
        depth, k, nargs, any_arg = 1, m.end(), 0, False
        while depth:                                   # the balanced argument list: top-level commas separate arguments
            ch = src[k]
            depth += ch in "([{"; depth -= ch in ")]}"
            if depth == 1 and ch == ",":
                nargs += 1
            any_arg = any_arg or (depth >= 1 and not ch.isspace() and ch != ")")
            k += 1
        calls.append((m.group(1), nargs + 1 if any_arg else 0))
    assert len(calls) > 20
    for name, na in calls:
        assert name in decl, name
        nd = 0 if decl[name].strip() in ("", "void") else decl[name].count(",") + 1
        assert na == nd, (name, na, nd)
    for need in ("sz_timestep_collisions", "sz_timestep_coupling", "sz_timestep_floe_properties", "sz_step", "sz_upload_interactions",
                 "sz_download_fuse", "sz_simplify_check", "sz_get_boundary_rects"):
        assert any(n == need for n, _ in calls), need


def test_hot_kernels_keep_their_register_budgets():
    """Occupancy is part of the measured performance and changes silently: the neighbour search with the family records sits
    exactly at the 168 registers that three wavefronts per SIMD allow (one more costs a third of its throughput: 17 -> 23 us at
    10 k floes), the lean one under the 128 of four, the narrow phase at 168 with 16 KB of LDS per wavefront (ten wavefronts per
    CU), and none of them may spill.  Read from the code object's metadata -- no GPU needed."""
    import importlib.util
    from subzero_jl_amd import build as _b
    _b.build()                                  # (no-op when the library is up to date)
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec); spec.loader.exec_module(kr)
    res = kr.resources()
    budget = {"sz_k_neighbors<true, 24, false>": (168, 0), "sz_k_neighbors<false, 24, false>": (128, 0), "sz_k_neighbors<true, 64, false>": (256, 0),      # (35 KB of LDS per two wavefronts: two per SIMD whatever the registers)
              # (the instantiations on collision records, State::crec -- what the resident steps run)
              "sz_k_neighbors<true, 24, true>": (168, 0), "sz_k_neighbors<false, 24, true>": (128, 0),
              # (the narrow phase parks the prefetched first work item -- four loop-invariant words -- in scratch once per launch: 16 bytes + one
              #  word, stored before the first round and read back by the round that uses it)
              "sz_k_narrow<8, 18, 8, 16, 4, 64, 0, 0, 3, 0, 0>": (168, 20), "sz_k_narrow<8, 18, 8, 16, 4, 64, 0, 0, 3, 1, 0>": (168, 20),
              # (pipelined steps: the same launch with GEO workgroups behind the narrow ones, and the update beside the neighbour search -- both at
              #  the three-wavefront budget; their rare paths (a parent that gets ghosts) may spill)
              "sz_k_narrow<8, 18, 8, 16, 4, 64, 0, 0, 3, 1, 1>": (168, 400), "sz_k_vel_search<true>": (168, 400),
              "sz_k_inter_fill": (128, 0), "sz_k_forcing<false>": (128, 0), "sz_k_forcing_mixed": (80, 16), "sz_k_halo_pack": (128, 0)}
    for name, (vg, scratch) in budget.items():
        assert name in res, (name, sorted(k for k in res if k.startswith(name.split("<")[0])))
        assert res[name]["vgpr"] <= vg and res[name]["scratch"] <= scratch, (name, res[name])
    assert res["sz_k_narrow<8, 18, 8, 16, 4, 64, 0, 0, 3, 1, 0>"]["lds"] <= 16384


def test_tile_fuzzer_seeds_keep_their_meaning():
    """tools/fuzz_tiles.py: the cases recorded under profiles/r03_runs/r3_t_fuzz_tiles.txt and kept as GPU tests are named by seed -- the derivation of a
    case from its seed must not drift"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_tiles", os.path.join(ROOT, "tools", "fuzz_tiles.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    assert m.case_params(5112) == (2, 573, 25, 10, "fast", "star", True, False)
    assert m.case_params(5090) == (4, 2336, 27, 9, "fast", "star", True, False)
    assert m.case_params(5056) == (2, 703, 54, 13, "fast", "star", True, False)
    assert m.case_params(30025, walls=True)[:5] == (4, 2006, 40, 9, "walls-topo-fast")
    assert m.case_params(30071, walls=True)[:5] == (4, 1943, 25, 12, "walls")
    w, n, steps, every, kind, shape, fast, stop = m.case_params(9003, mixed=True)
    assert kind == "voronoi-fast" and n == 3600 and w == 4 and shape == "voronoi" and fast and not stop
