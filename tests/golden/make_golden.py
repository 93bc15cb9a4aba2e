#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/.

Run in the authoring container only (it reads /root/reference, which does not exist on the GPU
box).  The fixtures are DATA: literal inputs and expected outputs transcribed from the
reference's own known-answer tests, plus the sub-floe points its forcing test loads from a
data file.  No reference source text is copied.

  collisions.json  <- test/test_physical_processes/test_collisions.jl:43-362 (literal inputs,
                      MATLAB-derived expected values, tolerances as the reference states them)
  floe_utils.json  <- test/test_floe_utils.jl:52-63 (translate), :66-71 (moment of inertia), :74-137
                      (which_vertices_match_points), :173-192 (rotate_radians!)
  boundaries.json  <- test/test_simulation_components/domain_components/boundaries.jl:5-127 (boundary rectangles,
                      _update_boundary!)
  conservation.json <- test/test_conservation.jl:58-203 (the energy / momentum conservation runs: three with literal floes, two with
                      the many-sided non-convex outlines floe_vertices[1, 3, 4, 5] of test/inputs/floe_shapes.jld2, decoded by
                      following the JLD2 container's nested arrays of object references -- _H5 below)
  forcings.json    <- test/test_physical_processes/test_coupling.jl:464-639, with the sub-floe
                      points X, Y decoded from test/inputs/test_mc_points.jld2 (two contiguous
                      little-endian Float64 datasets of 241 values each inside the JLD2/HDF5
                      container; located by scanning, since no HDF5 reader exists in the image)
  update_floe.json <- test/test_physical_processes/test_update_floe.jl:2-43 (expected stress / strain),
                      with the two floes of test/inputs/stress_strain.jld2 decoded from the container:
                      the root group's link messages give the object-header address of every named
                      dataset, the small Float64 datasets sit 61 bytes behind their header, and the
                      ring points are consecutive 2-value datasets (81 bytes apart) in ring order
  coupling_grid.json <- test/test_physical_processes/test_coupling.jl:165-180,276-460 (centre-cell index,
                      centre-cell rectangles, floe_to_grid_info! bookkeeping)
"""
import re
import struct
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

Lx = Ly = 1e5


def translate(c, dx, dy):
    return [[p[0] + dx, p[1] + dy] for p in c]


def collisions():
    corner_rect = [[0.0, 2.5e4], [0.0, 2.9e4], [2e4, 2.9e4], [2e4, 2.5e4], [0.0, 2.5e4]]
    g = {
        "_source": "test/test_physical_processes/test_collisions.jl",
        "grid": {"x0": -Lx, "xf": Lx, "y0": -Ly, "yf": Ly},          # :10-12
        "dt": 10, "hmean_floe_floe": 0.25, "max_overlap_floe_floe": 0.55,   # :3,40-41
        "coords": {                                                  # :43-48
            "tri": [[0.0, 0.0], [1e4, 3e4], [2e4, 0.0], [0.0, 0.0]],
            "corner_rect": corner_rect,
            "small_shift_corner_rect": translate(corner_rect, 0.5e4, 0.0),
            "big_shift_corner_rect": translate(corner_rect, 1.9999999e4, 0.0),
            "middle_rect": [[1.8e4, 2.7e4], [1.8e4, 2.8e4], [2.1e4, 2.8e4], [2.1e4, 2.7e4], [1.8e4, 2.7e4]],
            "cshape": [[0.5e4, 2.7e4], [0.5e4, 3.5e4], [1.5e4, 3.5e4], [1.5e4, 2.7e4], [1.25e4, 2.7e4],
                       [1.25e4, 3e4], [1e4, 3e4], [1e4, 2.7e4], [0.5e4, 2.7e4]],
        },
        "floe_floe": [
            {   # :51-62
                "name": "tri_tip_in_rect", "i": "tri", "j": "corner_rect",
                "vel_i": {"u": 0.1}, "vel_j": {"v": -0.1},
                "rows": [{"xforce": -64613382.47, "yforce": -521498991.51, "xpoint": 10000.00,
                          "ypoint": 26555.55, "overlap": 8000000.0, "torque": 1069710443203.99}],
                "atol": 1e-2, "fuse": False,
            },
            {   # :65-81
                "name": "cshape_two_regions", "i": "cshape", "j": "corner_rect",
                "vel_i": {"u": 0.3}, "vel_j": {"v": -0.1},
                "rows": [
                    {"xforce": -163013665.41, "yforce": 804819565.60, "xpoint": 7500.00, "ypoint": 28000.00,
                     "overlap": 10000000.0, "torque": -2439177121266.03},
                    {"xforce": -81506832.70, "yforce": 402409782.80, "xpoint": 13750.00, "ypoint": 28000.00,
                     "overlap": 5000000.0, "torque": 1295472581868.05},
                ],
                "atol": 1e-2, "fuse": False,
            },
            {"name": "fuse_small_shift", "i": "corner_rect", "j": "small_shift_corner_rect",     # :84-89
             "vel_i": {"v": -0.1}, "vel_j": {"v": -0.1}, "rows": [], "fuse": True},
            {"name": "fuse_middle_rect", "i": "corner_rect", "j": "middle_rect",                 # :92-96
             "vel_i": {"v": -0.1}, "vel_j": {}, "rows": [], "fuse": True},
            {"name": "tiny_overlap_no_force", "i": "big_shift_corner_rect", "j": "corner_rect",  # :99-102
             "vel_i": {"v": -0.1}, "vel_j": {"v": -0.1}, "rows": [], "fuse": False},
        ],
        "boundary": {   # :105-187
            "hmean": 0.25, "max_overlap": 0.75,
            # topo_domain: north/south periodic, east collision, west open, one topography element (:30-32)
            "topo_domain": {"kinds": ["periodic", "periodic", "collision", "open"],
                            "topography": [[[1e4, 0.0], [0.0, 1e4], [1e4, 2e4], [2e4, 1e4], [1e4, 0.0]]]},
            "collision_domain": {"kinds": ["collision"] * 4, "topography": []},
            "coords": {
                "north": [[5e4, 9.75e4], [5e4, 10.05e4], [7e4, 10.05e4], [7e4, 9.75e4], [5e4, 9.75e4]],
                "east_small": [[9.5e4, 0.0], [9e4, 0.5e4], [10e4, 2.5e4], [10.05e4, 2e4], [9.5e4, 0.0]],
                "east_large": [[9e4, -7e4], [9e4, -5e4], [1.4e5, -5e4], [1.4e5, -7e4], [9e4, -7e4]],
                "west": [[-9.75e4, 7e4], [-9.75e4, 5e4], [-10.05e4, 5e4], [-10.05e4, 7e4], [-9.75e4, 7e4]],
                "cshape": [[9.5e4, 7e4], [9.5e4, 9e4], [1.05e5, 9e4], [1.05e5, 8.5e4], [9.9e4, 8.5e4],
                           [9.9e4, 8e4], [1.05e5, 8e4], [1.05e5, 7e4], [9.5e4, 7e4]],
                "topo_overlap": [[-0.5e4, 0.0], [-0.5e4, 0.75e4], [0.5e4, 0.75e4], [0.5e4, 0.0], [-0.5e4, 0.0]],
                "corner": [[9.5e4, 7e4], [9e4, 7.5e4], [10e4, 1.05e5], [10.05e4, 9.5e4], [9.5e4, 7e4]],
            },
            "cases": [
                {"name": "east_small", "floe": "east_small", "domain": "topo_domain", "vel": {"u": 0.5, "v": 0.25},   # :125-133
                 "rows": [{"floeidx": -3, "xforce": -311304795.629, "yforce": -23618874.648,
                           "overlap": 1704545.454, "xpoint": 100166.666, "ypoint": 21060.606}], "atol": 1e-3},
                {"name": "cshape_two_regions", "floe": "cshape", "domain": "topo_domain", "vel": {"v": -0.1},        # :136-150
                 "rows": [{"floeidx": -3, "xforce": -2876118708.17, "yforce": 575223741.63, "xpoint": 102500.0,
                           "ypoint": 87500.0, "overlap": 25000000.0},
                          {"floeidx": -3, "xforce": -5752237416.35, "yforce": 1150447483.27, "xpoint": 102500.0,
                           "ypoint": 75000.0, "overlap": 50000000.0}], "atol": 1e-2},
                {"name": "east_large_removed", "floe": "east_large", "domain": "topo_domain",                        # :153-157
                 "vel": {"u": -0.4, "v": 0.2}, "rows": [], "status": "remove"},
                {"name": "east_large_maxoverlap1", "floe": "east_large", "domain": "topo_domain",                    # :160-164
                 "vel": {"u": -0.4, "v": 0.2}, "max_overlap": 1.0, "min_rows": 1},
                {"name": "west_open_removed", "floe": "west", "domain": "topo_domain", "vel": {},                    # :167-169
                 "rows": [], "status": "remove"},
                {"name": "north_periodic_noop", "floe": "north", "domain": "topo_domain", "vel": {}, "rows": []},    # :172-174
                {"name": "topography", "floe": "topo_overlap", "domain": "topo_domain", "vel": {},                   # :177-181
                 "first_row_floeidx": -5, "first_row_force_negative": True},
                {"name": "corner_two_walls", "floe": "corner", "domain": "collision_domain", "vel": {},              # :184-187
                 "all_forces_nonpositive": True},
            ],
        },
        "add_ghosts": {   # :190-258
            "hmean": 0.5,
            "coords": [
                [[9.9e4, 9.9e4], [9.9e4, 1.02e5], [1.02e5, 1.02e5], [1.02e5, 9.9e4], [9.9e4, 9.9e4]],
                [[-1.01e5, 7e4], [-1.01e5, 8e4], [-8e4, 8e4], [-8e4, 7e4], [-1.01e5, 7e4]],
                [[-2e4, 9.5e4], [-2e4, 1.1e5], [-1e4, 1.1e5], [-1e4, 9.5e4], [-2e4, 9.5e4]],
                [[0.0, 0.0], [0.0, 2e4], [2e4, 2e4], [2e4, 0.0], [0.0, 0.0]],
            ],
            "cases": [
                {"name": "open", "kinds": ["open"] * 4, "n": 4, "shifts": [[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0]],
                 "id": [1, 2, 3, 4], "ghost_id": [0, 0, 0, 0], "ghosts": [[], [], [], []]},
                # shifts: [source coords index, dx/Lbox, dy/Lbox] per floe, Lbox = 2e5
                {"name": "ew", "kinds": ["open", "open", "periodic", "periodic"], "n": 6,
                 "shifts": [[0, -1, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0], [0, 0, 0], [1, 1, 0]],
                 "id": [1, 2, 3, 4, 1, 2], "ghost_id": [0, 0, 0, 0, 1, 1],
                 "ghosts": [[5], [6], [], [], [], []]},
                {"name": "ns", "kinds": ["periodic", "periodic", "open", "open"], "n": 6,
                 "shifts": [[0, 0, -1], [1, 0, 0], [2, 0, -1], [3, 0, 0], [0, 0, 0], [2, 0, 0]],
                 "id": [1, 2, 3, 4, 1, 3], "ghost_id": [0, 0, 0, 0, 1, 1],
                 "ghosts": [[5], [], [6], [], [], []]},
                {"name": "double", "kinds": ["periodic"] * 4, "n": 9,
                 "shifts": [[0, -1, -1], [1, 0, 0], [2, 0, -1], [3, 0, 0], [0, 0, 0], [1, 1, 0], [0, 0, -1],
                            [0, -1, 0], [2, 0, 0]],
                 "id": [1, 2, 3, 4, 1, 2, 1, 1, 3], "ghost_id": [0, 0, 0, 0, 1, 1, 2, 3, 1],
                 "ghosts": [[5, 7, 8], [6], [9], [], [], [], [], [], []]},
            ],
        },
        "ghost_collisions": {   # :260-362 ; all on the doubly periodic domain, hmean 0.5
            "hmean": 0.5,
            "lshape": [[Lx / 2, Ly / 2], [Lx / 2, Ly + 10000], [3 * Lx / 4, Ly + 10000], [3 * Lx / 4, 3 * Ly / 4],
                       [Lx + 10000, 3 * Ly / 4], [Lx + 10000, Ly / 2]],
            "oval": {"r": Ly / 4 + 1000, "cx": Lx - 1, "cy": Ly - 1, "nth": 101},   # th = 0:pi/50:2pi
            "tall_rect": [[5 * Lx / 8 + 1000, 3 * Ly / 4], [5 * Lx / 8 + 1000, 5 * Ly / 4],
                          [3 * Lx / 4 + 1000, 5 * Ly / 4], [3 * Lx / 4 + 1000, 3 * Ly / 4]],
            "long_rect": [[-5 * Lx / 4, -7 * Lx / 8], [-5 * Lx / 4, -(3 * Lx / 4 - 1000)],
                          [-(3 * Lx / 4 - 1000), -(3 * Lx / 4 - 1000)], [-(3 * Lx / 4 - 1000), -7 * Lx / 8]],
            "shifted_up_long_rect_dy": 1.615 * Ly,
            "small_corner_rect": [[-1.1e5, -1.1e5], [-1.1e5, -9.5e4], [-9.5e4, -9.5e4], [-9.5e4, -1.1e5], [-1.1e5, -1.1e5]],
            "large_tri": [[-1e5, -1e5], [-1e5, 1e5], [1e5, -1e5], [-1e5, -1e5]],
            "south_bound_rect": [[-9.8e4, -1.1e5], [-9.8e4, -9.5e4], [9.8e4, -9.5e4], [9.8e4, -1.1e5], [-9.8e4, -1.1e5]],
        },
    }
    return g


def floe_utils():
    ext = [[0.0, 1.0], [0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 1.0]]
    r2 = 2.0 ** 0.5
    return {
        "_source": "test/test_floe_utils.jl:12-14,52-63,66-71,74-137,173-192",
        "moment": [
            {"coords": ext, "height": 0.25, "expected": 38.333, "atol": 1e-3},
            {"coords": [[0.0, 6.67], [0.0, 0.0], [6.67, 0.0], [0.0, 6.67]], "height": 0.5,
             "expected": 50581.145, "atol": 1e-3},
        ],
        # :52-63  translate / translate!: exact equality in the reference
        "translate": [
            {"coords": ext, "dx": 0.0, "dy": 0.0, "expected": ext},
            {"coords": ext, "dx": 1.0, "dy": 2.0, "expected": [[1.0, 3.0], [1.0, 2.0], [2.0, 2.0], [2.0, 3.0], [1.0, 3.0]]},
            {"coords": [[-2.0, 2.0], [-2.0, 1.0], [-1.0, 1.0], [-1.0, 2.0]], "dx": 1.5, "dy": -1.5,
             "expected": [[-0.5, 0.5], [-0.5, -0.5], [0.5, -0.5], [0.5, 0.5]]},
        ],
        # :173-192  rotate_radians! about the origin, isapprox (rtol sqrt(eps)) in the reference
        "rotate": [
            {"coords": [[-1.0, -1.0], [-1.0, 1.0], [1.0, 1.0], [1.0, -1.0], [-1.0, -1.0]], "angle": "pi/4",
             "expected": [[0.0, -r2], [-r2, 0.0], [0.0, r2], [r2, 0.0], [0.0, -r2]]},
            {"coords": [[0.0, -r2], [-r2, 0.0], [0.0, r2], [r2, 0.0], [0.0, -r2]], "angle": "7pi/4",
             "expected": [[-1.0, -1.0], [-1.0, 1.0], [1.0, 1.0], [1.0, -1.0], [-1.0, -1.0]]},
        ],
        # :74-137  which_vertices_match_points(points = first ring, region = second polygon): 1-based vertex indices
        "which_vertices_match_points": [
            {"name": "two_shared_v",
             "points": [[0.0, 0.0], [0.0, 20.0], [20.0, 20.0], [20.0, 0.0], [0.0, 0.0]],
             "region": [[20.0, 0.0], [20.0, 20.0], [40.0, 20.0], [40.0, 0.0], [20.0, 0.0]], "expected": [1, 2]},
            {"name": "three_shared_v",
             "points": [[0.0, 0.0], [0.0, 20.0], [20.0, 20.0], [20.0, 10.0], [20.0, 0.0], [0.0, 0.0]],
             "region": [[40.0, 20.0], [40.0, 0.0], [20.0, 0.0], [20.0, 10.0], [20.0, 20.0], [40.0, 20.0]], "expected": [3, 4, 5]},
            {"name": "four_shared_v",
             "points": [[0.0, 0.0], [0.0, 20.0], [20.0, 20.0], [20.0, 18.0], [20.0, 15.0], [20.0, 0.0], [0.0, 0.0]],
             "region": [[20.0, 18.0], [20.0, 20.0], [40.0, 20.0], [40.0, 0.0], [20.0, 0.0], [20.0, 15.0], [20.0, 18.0]],
             "expected": [1, 2, 5, 6]},
            {"name": "triange_shared_v",
             "points": [[0.0, 0.0], [0.0, 20.0], [20.0, 20.0], [5.0, 5.0], [0.0, 0.0]],
             "region": [[0.0, 0.0], [5.0, 5.0], [20.0, 20.0], [20.0, 0.0], [0.0, 0.0]], "expected": [1, 2, 3]},
        ],
    }


def boundaries():
    """test/test_simulation_components/domain_components/boundaries.jl: the boundary rectangles of an extent (:5-31,
    :33-83 -- corner point sets and wall values), and _update_boundary! (:103-127: only MovingBoundary moves, the
    southern one by dt * v in y, the western one by dt * u in x; val follows).  The reference's western wall is a
    Float32 boundary with u = 0.1f0; the fixture states the same case in Float64."""
    return {
        "_source": "test/test_simulation_components/domain_components/boundaries.jl:5-31,33-83,90-127",
        "directions": {"extent": [0.0, 1e5, -5e4, 5e4],         # x0, xf, y0, yf
                       "north": {"val": 5e4, "points": [[-5e4, 5e4], [-5e4, 1e5], [1.5e5, 1e5], [1.5e5, 5e4]]},
                       "south": {"val": -5e4, "points": [[-5e4, -5e4], [-5e4, -1e5], [1.5e5, -1e5], [1.5e5, -5e4]]},
                       "east": {"val": 1e5, "points": [[1e5, -1e5], [1e5, 1e5], [1.5e5, -1e5], [1.5e5, 1e5]]},
                       "west": {"val": 0.0, "points": [[0.0, -1e5], [0.0, 1e5], [-5e4, -1e5], [-5e4, 1e5]]}},
        "boundaries": {"extent": [0.0, 4e5, 0.0, 3e5],
                       "north": {"val": 3e5, "area": 8e5 * 1.5e5, "points": [[-2e5, 3e5], [-2e5, 4.5e5], [6e5, 4.5e5], [6e5, 3e5]]},
                       "east": {"val": 4e5, "area": 2e5 * 6e5, "points": [[4e5, -1.5e5], [4e5, 4.5e5], [6e5, 4.5e5], [6e5, -1.5e5]]},
                       "south": {"val": 0.0, "area": 8e5 * 1.5e5, "points": [[-2e5, -1.5e5], [-2e5, 0.0], [6e5, 0.0], [6e5, -1.5e5]]},
                       "west": {"val": 0.0, "area": 2e5 * 6e5, "points": [[-2e5, -1.5e5], [-2e5, 4.5e5], [0.0, 4.5e5], [0.0, -1.5e5]]}},
        # _update_boundary!(b, dt = 20): kinds N, S, E, W = open, moving (u 1, v 2), collision, moving (u 0.1, v -0.1)
        "update": {"extent": [0.0, 4e5, 0.0, 3e5], "dt": 20, "kinds": ["open", "moving", "collision", "moving"],
                   "u": [0.0, 1.0, 0.0, 0.1], "v": [0.0, 2.0, 0.0, -0.1],
                   "expected_vals": [3e5, 0.0 + 20 * 2.0, 4e5, 0.0 + 20 * 0.1],
                   "expected_shift": [[0.0, 0.0], [0.0, 20 * 2.0], [0.0, 0.0], [20 * 0.1, 0.0]]},
    }


class _H5:
    """Just enough of HDF5 to follow JLD2's nested arrays of references (no HDF5 reader exists in the image): version-2 object
    headers (OHDR, continuation blocks OCHK), the dataspace (0x01), datatype (0x03), data layout (0x08: compact or contiguous)
    and filter pipeline (0x0B: none of the datasets read here has one -- asserted) messages.  Addresses are relative to the
    superblock, which JLD2 puts behind a 512-byte user block."""

    def __init__(self, path):
        self.b = open(path, "rb").read(); self.base = 512
        assert self.b[self.base:self.base + 8] == b"\x89HDF\r\n\x1a\n" and self.b[self.base + 8] == 2      # superblock version 2
        self.root = struct.unpack_from("<Q", self.b, self.base + 12 + 24)[0]

    def _msgs(self, p, end, fl, out):
        b = self.b
        while p + 4 <= end:
            t = b[p]; sz = struct.unpack_from("<H", b, p + 1)[0]; p += 4
            if fl & 0x04:
                p += 2
            if t == 0x10:             # continuation: (address, length) of an OCHK block
                off, ln = struct.unpack_from("<QQ", b, p)
                q = self.base + off
                assert b[q:q + 4] == b"OCHK"
                self._msgs(q + 4, q + ln - 4, fl, out)
            elif t != 0:
                out.append((t, p, sz))
            p += sz

    def messages(self, rel):
        b = self.b; o = self.base + rel
        assert b[o:o + 4] == b"OHDR" and b[o + 4] == 2, rel
        fl = b[o + 5]; p = o + 6 + (16 if fl & 0x20 else 0) + (4 if fl & 0x10 else 0)
        szb = 1 << (fl & 3)
        size = int.from_bytes(b[p:p + szb], "little"); p += szb
        out = []
        self._msgs(p, p + size, fl, out)
        return out

    def links(self, rel):
        """name -> object header address of a group's link messages (version 1, hard links, 1-byte name length)"""
        out = {}
        for t, p, sz in self.messages(rel):
            if t == 0x06:
                d = self.b[p:p + sz]
                assert d[0] == 1 and d[1] == 0x10 and d[2] == 1, d[:4]
                n = d[3]
                out[d[4:4 + n].decode("utf8")] = struct.unpack_from("<Q", d, 4 + n)[0]
        return out

    def dataset(self, rel):
        """(datatype class, element size, dims, raw bytes) of a dataset"""
        cls = size = dims = raw = None
        for t, p, sz in self.messages(rel):
            d = self.b[p:p + sz]
            if t == 0x01:
                rank = d[1]; dims = struct.unpack_from("<%dQ" % rank, d, 4 if d[0] == 2 else 8)
            elif t == 0x03:
                cls = d[0] & 0xf; size = struct.unpack_from("<I", d, 4)[0]
            elif t == 0x08:
                if d[1] == 0:
                    n = struct.unpack_from("<H", d, 2)[0]; raw = self.b[p + 4:p + 4 + n]
                elif d[1] == 1:
                    addr, n = struct.unpack_from("<QQ", d, 2); raw = self.b[self.base + addr:self.base + addr + n]
                else:
                    raise AssertionError("chunked dataset: not expected in these files")
            elif t == 0x0B:
                raise AssertionError("filtered (compressed) dataset: no codec in the image")
        return cls, size, dims, raw


def decode_floe_shapes(which):
    """file["floe_vertices"][k] for the 1-based k in `which` out of test/inputs/floe_shapes.jld2: a 462 x 1 array of references,
    each a PolyVec = Vector (rings) of Vector (points) of Vector{Float64}(2) -- every level an array of 8-byte references to the
    next dataset (datatype class 7), the points little-endian Float64 pairs (class 1, size 8).  Returns {k: [ring, ..]}."""
    h = _H5(os.path.join(REF, "test/inputs/floe_shapes.jld2"))
    cls, size, dims, raw = h.dataset(h.links(h.root)["floe_vertices"])
    assert cls == 7 and size == 8 and tuple(dims) == (1, 462), (cls, size, dims)
    top = struct.unpack("<462Q", raw)
    out = {}
    for k in which:
        cls, size, dims, raw = h.dataset(top[k - 1])
        assert cls == 7 and size == 8
        rings = []
        for rr in struct.unpack("<%dQ" % (len(raw) // 8), raw):
            cls2, size2, dims2, raw2 = h.dataset(rr)
            assert cls2 == 7 and size2 == 8
            pts = []
            for pr in struct.unpack("<%dQ" % (len(raw2) // 8), raw2):
                cls3, size3, dims3, raw3 = h.dataset(pr)
                assert cls3 == 1 and size3 == 8 and tuple(dims3) == (2,), (cls3, size3, dims3)
                pts.append(list(struct.unpack("<2d", raw3)))
            rings.append(pts)
        assert len(rings) == 1            # no holes
        out[k] = rings
    return out


def _valid_ring(r):
    """valid_ringvec! (floe_utils.jl:10-17): adjacent duplicates dropped, ring closed"""
    r = [list(p) for p in r]
    r = [p for i, p in enumerate(r) if i == len(r) - 1 or p != r[i + 1]]
    if r[0] != r[-1]:
        r.append(list(r[0]))
    return r


def conservation():
    """test/test_conservation.jl:58-203: the three runs with literal floes (two blocks head on, offset, and with a
    triangle between them), the run with three many-sided non-convex floes (:156-182, criterion 2.1 %) and the run of one
    non-convex floe next to a wall and a topography element (:184-203, energy only).  dt = 1 s, 5000 steps,
    E = 1.5e3 (mean sqrt(area) + min sqrt(area)), mu = 0, coupling off, open domain, hmean 0.25 (:1-56); pass = |change| of
    total kinetic energy, x momentum, y momentum and total angular momentum (src/tools/conservation_em.jl:16-67) from the
    first to the last output below the criterion.  The complex shapes are floe_vertices[1, 3, 4, 5] of
    test/inputs/floe_shapes.jld2 (decode_floe_shapes: rings of 35 / 50 / 146 / 203 points once closed), translated as the test
    does (Subzero.translate); none of them overlaps another floe or the topography at the start (so the diff_polys of
    initialize_floe_field, floe.jl:376-378, leaves them as they are)."""
    shp = decode_floe_shapes([1, 3, 4, 5])
    assert [len(_valid_ring(shp[k][0])) for k in (1, 3, 4, 5)] == [35, 50, 146, 203]
    floe1 = [[2e4, 2e4], [2e4, 5e4], [5e4, 5e4], [5e4, 2e4], [2e4, 2e4]]
    floe2 = [[6e4, 2e4], [6e4, 5e4], [9e4, 5e4], [9e4, 2e4], [6e4, 2e4]]
    floe3 = [[5.5e4, 2e4], [5.25e4, 4e4], [5.75e4, 4e4], [5.5e4, 2e4]]
    return {
        "_source": "test/test_conservation.jl:1-146, src/tools/conservation_em.jl:16-67",
        "grid": {"x0": -2e4, "xf": 1e5, "y0": 0.0, "yf": 1e5, "dx": 1e4, "dy": 1e4},
        "dt": 1, "nsteps": 5000, "mu": 0.0, "hmean": 0.25, "boundaries": "open", "max_percent_change": 1.0,
        "cases": [
            {"name": "head_on", "floes": [floe1, floe2], "u": [0.15, -0.1], "v": [0.02, 0.02], "xi": [1e-7, 0.0]},
            {"name": "offset", "floes": [floe1, translate(floe2, 0.0, 1e4)], "u": [0.11, -0.1], "v": [0.02, 0.02], "xi": [1e-7, 0.0]},
            {"name": "rotating", "floes": [floe1, floe2, floe3], "u": [0.11, -0.1, 0.0], "v": [0.001, 0.001, 0.001],
             "xi": [0.0, 0.0, 1e-5]},
            {"name": "complex_shapes", "floes": [translate(_valid_ring(shp[3][0]), 0.0, 2e4), _valid_ring(shp[4][0]), _valid_ring(shp[5][0])],
             "u": [0.1, 0.0, 0.0], "v": [0.0, -0.2, 0.2], "xi": [0.0, 0.0, 0.0], "max_percent_change": 2.1,
             "_source": "test/test_conservation.jl:156-182"},
            {"name": "wall_and_topography", "floes": [translate(_valid_ring(shp[1][0]), -1.75e4, -0.9e4)],
             "u": [-0.09], "v": [-0.09], "xi": [0.0], "max_percent_change": 1.0, "only": [0],
             "topography": [[[-1e4, 0.0], [-2e4, 1e4], [-1e4, 1e4], [-1e4, 0.0]]],
             "_source": "test/test_conservation.jl:184-203 (energy only)"},
        ],
    }


def decode_mc_points():
    b = open(os.path.join(REF, "test/inputs/test_mc_points.jld2"), "rb").read()
    runs = []
    for off in range(8):
        a = np.frombuffer(b[off:off + ((len(b) - off) // 8) * 8], dtype="<f8")
        ok = np.isfinite(a) & (np.abs(a) > 1e-3) & (np.abs(a) < 1e6)
        s = None
        for i, v in enumerate(ok):
            if v and s is None:
                s = i
            if not v and s is not None:
                if i - s > 20:
                    runs.append((off + 8 * s, i - s))
                s = None
        if s is not None and len(ok) - s > 20:
            runs.append((off + 8 * s, len(ok) - s))
    runs.sort()
    assert len(runs) == 2 and runs[0][1] == runs[1][1] == 241, runs
    arrs = [np.frombuffer(b[o:o + 8 * n], dtype="<f8") for o, n in runs]
    # the floe is 5 km wide and 20 km tall, centred at the origin: X is the narrow one
    X, Y = (arrs if np.abs(arrs[0]).max() < np.abs(arrs[1]).max() else arrs[::-1])
    assert np.abs(X).max() <= 2500.0 and np.abs(Y).max() <= 10000.0
    return X.tolist(), Y.tolist()


def forcings():
    X, Y = decode_mc_points()
    return {
        "_source": "test/test_physical_processes/test_coupling.jl:464-639; X,Y from test/inputs/test_mc_points.jld2",
        "grid": {"x0": -1e5, "xf": 1e5, "y0": -1e5, "yf": 1e5, "dx": 1e4, "dy": 1e4},
        "domain_kinds": ["collision"] * 4,
        "floe": [[-1.75e4, 5e4], [-1.75e4, 7e4], [-1.25e4, 7e4], [-1.25e4, 5e4], [-1.75e4, 5e4]],
        "height": 0.25, "X": X, "Y": Y,
        "psi": "0.5e4*sin(4*(pi/4e5)*x)*sin(4*(pi/4e5)*y) on the grid-line lattice (:577-592)",
        "cases": [
            {"name": "zonal_ocean", "ocean": [1.0, 0.0], "atmos": [0.0, 0.0], "floe_uv": [0.0, 0.0], "dd": 2,
             "fx": 2.9760, "fy": 0.8296, "trq": -523.9212, "atol": [1e-3, 1e-3, 1e-3]},
            {"name": "meridional_ocean", "ocean": [0.0, 1.0], "atmos": [0.0, 0.0], "floe_uv": [0.0, 0.0], "dd": 2,
             "fx": -0.8296, "fy": 2.9760, "trq": 239.3141, "atol": [1e-3, 1e-3, 1e-3]},
            {"name": "moving_floe", "ocean": [0.0, 0.0], "atmos": [0.0, 0.0], "floe_uv": [0.25, 0.1], "dd": 2,
             "fx": -0.1756, "fy": -0.1419, "trq": 29.0465, "atol": [1e-3, 1e-3, 1e-1]},
            {"name": "diagonal_atmos", "ocean": [0.0, 0.0], "atmos": [-1.0, -0.5], "floe_uv": [0.0, 0.0], "dd": 2,
             "fx": -0.0013, "fy": -6.7082e-4, "trq": 0.2276, "atol": [1e-3, 1e-3, 1e-3]},
            {"name": "nonuniform_ocean", "ocean": "psi", "atmos": [0.0, 0.0], "floe_uv": [0.0, 0.0], "dd": 1,
             "fx": -0.0182, "fy": 0.0392, "trq": 23.6399, "atol": [1e-3, 1e-3, 1e-3]},
            {"name": "nonuniform_both_moving", "ocean": "psi", "atmos": "psi", "floe_uv": [0.5, -0.5], "dd": 1,
             "fx": -1.6300, "fy": 1.1240, "trq": 523.2361, "atol": [1e-3, 1e-3, 2e-1]},
        ],
    }


def decode_stress_strain():
    b = open(os.path.join(REF, "test/inputs/stress_strain.jld2"), "rb").read()
    base = 512                                           # JLD2 superblock offset: addresses are relative to it

    def dbl(o, n):
        return list(struct.unpack_from("<%dd" % n, b, o))

    # root group link messages: version 1, flags 0x10, charset 1, name length, name, address
    links = {}
    for m in re.finditer(rb"\x01\x10\x01(.)", b[8800:], re.S):
        ln = m.group(1)[0]; st = 8800 + m.end()
        links[b[st:st + ln].decode("utf8")] = base + struct.unpack_from("<Q", b, st + ln)[0]
    assert {"u", "v", "ξ", "height", "area", "interactions", "coords", "last_stress"} <= set(links), links
    small = {k: dbl(links[k] + 61, 2) for k in ("u", "v", "ξ", "height", "area")}      # Vector{Float64}(2) each
    assert small["height"] == [0.25, 0.25] and small["area"] == [8e7, 7.25e7], small

    def runs(lo, hi, n, pred):
        out = []
        o = lo
        while o + 8 * n <= hi:
            v = dbl(o, n)
            if pred(v):
                out.append((o, v)); o += 8 * n
            else:
                o += 1
        return out

    def plausible(v):
        return all(np.isfinite(x) and (x == 0.0 or 1e-3 < abs(x) < 1e13) for x in v)
    # interactions: two 2x7 matrices (column-major, 14 doubles) whose first column is the partner index
    inter = [v for _, v in runs(links["interactions"], links["coords"], 14,
                                lambda v: plausible(v) and v[0] == v[1] and v[0] in (1.0, 2.0) and abs(v[2]) > 1e6)]
    assert len(inter) == 2, inter
    inter = [np.array(v).reshape(7, 2).T.tolist() for v in inter]
    # ring points: 2-value datasets 81 bytes apart, in ring order; a ring ends when its first point recurs
    pts = runs(links["coords"], links["centroid"], 2,
               lambda v: plausible(v) and all(x == 0.0 or 1e3 <= abs(x) <= 1e5 for x in v) and any(x != 0.0 for x in v))
    pts = [(o, v) for o, v in pts if not (abs(v[0] - 1.2353074274786685e-4) < 1e-12)]
    rings, cur = [], []
    for k, (o, v) in enumerate(pts):
        if cur and o - pts[k - 1][0] != 81 and len(cur) < 4:
            cur = []
        cur.append(v)
        if len(cur) >= 4 and cur[-1] == cur[0]:
            rings.append(cur); cur = []
    assert len(rings) == 2 and len(rings[0]) == 5 and len(rings[1]) == 9, rings
    last = [v for _, v in runs(links["last_stress"], links["last_stress"] + 400, 4,
                               lambda v: plausible(v) and v[1] == v[2] and abs(v[0]) > 1e3)]
    assert len(last) == 2, last
    return small, inter, rings, last


def update_floe():
    small, inter, rings, last = decode_stress_strain()
    return {
        "_source": "test/test_physical_processes/test_update_floe.jl:2-43; floes from test/inputs/stress_strain.jld2",
        "dt": 10,
        "floes": [{"coords": rings[i], "height": small["height"][i], "u": small["u"][i], "v": small["v"][i],
                   "xi": small["ξ"][i], "area": small["area"][i], "interactions": inter[i],
                   "last_stress": last[i]} for i in range(2)],
        # calc_stress!: floe.stress_instant (the reference's variable name for it is stress_histories); the
        # stress_accum check is @test_broken in the reference and is not transcribed
        "stress_instant": [[-4971.252, 17483.052, 17483.052, -57097.458], [4028.520, 9502.886, 9502.886, -205199.791]],
        "strain_times_1e6": [[-0.0372, 0, 0, 0.9310], [7.419, 0, 0, -6.987]],
        "atol": 1e-3,
    }


def coupling_grid():
    """Grid bookkeeping of the coupling: test_coupling.jl:165-180 (find_center_cell_index), :181-197 (in_bounds), :199-282
    (find_interp_knots), :276-289 (center_cell_coords) and :291-460 (floe_to_grid_info!), literal inputs and expected values."""
    P, O = "periodic", "open"
    return {
        "_source": "test/test_physical_processes/test_coupling.jl:165-460",
        "grid": {"x0": -10.0, "xf": 10.0, "y0": -8.0, "yf": 8.0, "dx": 2.0, "dy": 4.0},
        "find_center_cell_index": {"x": [-10.5, -10, -10, -6.5, -6, -4, 10, 10.5, 12], "y": [0.0, 6.0, -8.0, 4.5, 0.0, 5.0, -8.0, 0.0, 0.0],
                                   "xidx": [1, 1, 1, 3, 3, 4, 11, 11, 12], "yidx": [3, 5, 1, 4, 3, 4, 1, 3, 3]},
        # (xidx, yidx, north/south kind, east/west kind) -> xmin, xmax, ymin, ymax of the expected rectangle
        "center_cell_coords": [
            {"idx": [2, 3], "ns": P, "ew": P, "rect": [-9, -7, -2, 2]},
            {"idx": [1, 1], "ns": O, "ew": O, "rect": [-10, -9, -8, -6]},
            {"idx": [11, 6], "ns": P, "ew": P, "rect": [9, 11, 10, 14]},
            {"idx": [11, 6], "ns": O, "ew": O, "rect": [9, 10, 8, 8]},
            {"idx": [11, 6], "ns": O, "ew": P, "rect": [9, 11, 8, 8]},
            {"idx": [11, 6], "ns": P, "ew": O, "rect": [9, 10, 10, 14]},
        ],
        # in_bounds(x, y, grid, north-south boundary, east-west boundary): test_coupling.jl:181-197 (source coupling.jl:494-597).  Keys name the
        # (north/south, east/west) kinds of the argument order of the reference's call.
        "in_bounds": {"x": [-12, -10, -8, -6, 0, 4, 4, 10, 12, 12], "y": [5, -6, 4, 10, -10, 8, -8, -6, 4, 10],
                      "open_open": [False, True, True, False, False, True, True, True, False, False],
                      "periodic_open": [False, True, True, True, True, True, True, True, False, False],
                      "open_periodic": [True, True, True, False, False, True, True, True, True, False],
                      "periodic_periodic": [True] * 10},
        # find_interp_knots(point_idx, ncells, glines, L, dd, boundary) -> (knots, knot_idx): test_coupling.jl:199-282 (source coupling.jl:702-797);
        # glines 0:10:80 (8 cells, L = 80); knot_idx 1-based grid-line numbers -- periodic: line ncells + 1 IS line 1 and is not repeated
        "find_interp_knots": [
            {"points": [4], "dd": 2, "periodic": False, "knots": list(range(0, 70, 10)), "idx": list(range(1, 8))},
            {"points": [4], "dd": 2, "periodic": True, "knots": list(range(0, 70, 10)), "idx": list(range(1, 8))},
            {"points": [0, 1], "dd": 2, "periodic": False, "knots": list(range(0, 40, 10)), "idx": list(range(1, 5))},
            {"points": [0, 1], "dd": 2, "periodic": True, "knots": list(range(-40, 40, 10)), "idx": [5, 6, 7, 8, 1, 2, 3, 4]},
            {"points": [8, 9], "dd": 1, "periodic": False, "knots": list(range(50, 90, 10)), "idx": list(range(6, 10))},
            {"points": [8, 9], "dd": 1, "periodic": True, "knots": list(range(50, 110, 10)), "idx": [6, 7, 8, 1, 2, 3]},
            {"points": list(range(1, 9)), "dd": 2, "periodic": False, "knots": list(range(0, 90, 10)), "idx": list(range(1, 10))},
            {"points": list(range(1, 9)), "dd": 2, "periodic": True, "knots": list(range(-30, 110, 10)), "idx": [6, 7, 8] + list(range(1, 9)) + [1, 2, 3]},
            {"points": list(range(0, 10)), "dd": 2, "periodic": False, "knots": list(range(0, 90, 10)), "idx": list(range(1, 10))},
        ],
        "knots_grid": {"g0": 0.0, "dg": 10.0, "ncells": 8, "L": 80.0},
        # floe_to_grid_info!(floeidx, xidx[i], yidx[i], tx[i], ty[i], grid, ns, ew, scells, two_way_coupling_on)
        "floe_to_grid": [
            {"floeidx": 1, "xidx": [7, 7, 6, 6, 7, 7], "yidx": [4, 4, 3, 3, 4, 4], "tx": 1.0, "ty": 2.0, "ns": O, "ew": O,
             "cells": [[7, 4], [6, 3]], "dx": [0.0, 0.0], "dy": [0.0, 0.0], "sum_tx": [-4, -2], "sum_ty": [-8, -4], "npoints": [4, 2]},
            {"floeidx": 2, "xidx": [7, 7, 8, 8, 9, 9], "yidx": [2, 3, 3, 3, 2, 2], "tx": 1.0, "ty": 2.0, "ns": P, "ew": P,
             "cells": [[7, 2], [7, 3], [9, 2], [8, 3]], "dx": [0.0] * 4, "dy": [0.0] * 4,
             "sum_tx": [-1, -1, -2, -2], "sum_ty": [-2, -2, -4, -4], "npoints": [1, 1, 2, 2]},
            {"floeidx": 3, "xidx": [10, 10, 10, 11, 11, 11, 11], "yidx": [4, 5, 6, 5, 6, 5, 6], "tx": 1.0, "ty": 2.0, "ns": P, "ew": O,
             "cells": [[10, 1], [11, 1], [10, 2], [11, 2], [10, 4]], "dx": [0.0] * 5, "dy": [-16.0, -16.0, -16.0, -16.0, 0.0],
             "sum_tx": [-1, -2, -1, -2, -1], "sum_ty": [-2, -4, -2, -4, -2], "npoints": [1, 2, 1, 2, 1]},
            {"floeidx": 4, "xidx": [11, 11, 12, 12, 11], "yidx": [4, 5, 5, 5, 4], "tx": 1.0, "ty": 2.0, "ns": O, "ew": P,
             "cells": [[1, 4], [1, 5], [2, 5]], "dx": [-20.0] * 3, "dy": [0.0] * 3,
             "sum_tx": [-2, -1, -2], "sum_ty": [-4, -2, -4], "npoints": [2, 1, 2]},
            {"floeidx": 2, "xidx": [0, -1, -1, 1, -1], "yidx": [0, -1, -2, 1, -1], "tx": -1.0, "ty": -2.0, "ns": P, "ew": P,
             "cells": [[1, 1], [10, 4], [9, 3], [9, 2]], "dx": [0.0, 20.0, 20.0, 20.0], "dy": [0.0, 16.0, 16.0, 16.0],
             "sum_tx": [1, 1, 2, 1], "sum_ty": [2, 2, 4, 2], "npoints": [1, 1, 2, 1]},
        ],
    }


def main():
    for name, fn in (("collisions.json", collisions), ("floe_utils.json", floe_utils), ("forcings.json", forcings),
                     ("update_floe.json", update_floe), ("coupling_grid.json", coupling_grid), ("boundaries.json", boundaries),
                     ("conservation.json", conservation)):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(fn(), f, indent=1)
        print("wrote", name)


if __name__ == "__main__":
    main()
