"""Rough test shapes: random star polygons with 3 to 40 vertices and deep radial variation on a jittered lattice
(used by tools/fuzz_parity.py and tests/test_hip_parity.py::test_rough_shapes)."""
import numpy as np


def rough_world(mk, seed, n_side, periodic):
    rng = np.random.default_rng(seed)
    L = 1e5; sp = L / n_side
    w = mk()
    w.set_consts(E=6e6); w.set_settings()
    kind = 1 if periodic else 2
    w.set_domain([kind] * 4, 0.0, L, 0.0, L)
    for gy in range(n_side):
        for gx in range(n_side):
            nv = int(rng.integers(3, 41))
            th = np.sort(rng.uniform(0, 2 * np.pi, nv))[::-1]
            if np.max(np.diff(np.concatenate([th[::-1], [th[-1] + 2 * np.pi]]))) >= np.pi * 0.95:
                th = (2 * np.pi / nv) * (np.arange(nv) + rng.uniform(-0.3, 0.3, nv))[::-1]
            rad = 0.75 * sp * rng.uniform(0.3, 1.0, nv)
            cx = (gx + 0.5) * sp + rng.uniform(-0.15, 0.15) * sp; cy = (gy + 0.5) * sp + rng.uniform(-0.15, 0.15) * sp
            ring = np.stack([cx + rad * np.cos(th), cy + rad * np.sin(th)], 1)
            w.add_floe(np.vstack([ring, ring[:1]]), 0.3)
    M = n_side * n_side
    w.set("u", rng.uniform(-0.2, 0.2, M)); w.set("v", rng.uniform(-0.2, 0.2, M)); w.set("xi", rng.uniform(-1e-5, 1e-5, M))
    return w
