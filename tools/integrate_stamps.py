"""Diagnostic: -DSZ_STAMPS build, clock of the thread that updates one floe (default 99: next to the E wall of the bench field, so
it makes a ghost) at a few points of sz_k_integrate<true>, in shader cycles since the thread's first stamp."""
import os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from subzero_jl_amd import build as b
lib = os.path.join(ROOT, "subzero.jl_amd", "libsubzero_hip_stamps.so")
floe = int(sys.argv[2]) if len(sys.argv) > 2 else 99
subprocess.check_call([b.hipcc()] + b.FLAGS + ["-DSZ_STAMPS", f"-DSZ_ISTAMP_FLOE={floe}", "-o", lib, b.SRC])
b.LIB = lib
import subzero_jl_amd
from subzero_jl_amd import fields, capi
capi._LIB = None
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfg = fields.make_config(n_floes=n, seed=12345)
names = ["thread starts its floe", "columns + ring offsets asked for, stop test passed", "ring + rows back, stress done", "force part done",
         "own stores issued", "ghost row built (ring moved again)", "ghosts made (allocation, cells, copies)"]
w = fields.build_world(subzero_jl_amd.World(0), cfg)
w.run(30, 0, cfg["dt"], coupling_dt=1)
out = np.zeros(512 + 8 * 8000, np.int64)
w._chk(w.L.sz_debug_stamps(w.h, capi.ptr(out, capi._lp)))
for rep in range(3):
    w.run(2, 30 + 2 * rep, cfg["dt"], coupling_dt=1)          # (the last step of a batch makes no ghosts: the stamps are the first step's)
    w._chk(w.L.sz_debug_stamps(w.h, capi.ptr(out, capi._lp)))
    t = out[900:907]
    print(f"floe {floe}:", "  ".join(f"{names[k]}: +{int(t[k] - t[k - 1]) if k and t[k] and t[k - 1] else 0}" for k in range(7)))
    g = out[905:916]           # inside ghost_inline_make: 5 entry | 10 plan done, cell counters next | 11 counters asked | 12 allocation back | 13 | 14 first copy stored | 15 cells placed
    if g[0]:
        print("   ghost_inline_make:", "  ".join(f"{lab} +{int(g[k] - g[j])}" for lab, j, k in (("plan (allocation asked, translations, cells)", 0, 5), ("cell counters asked", 5, 6),
              ("allocation back", 6, 7), ("checks", 7, 8), ("cell entries placed (counters back)", 8, 9), ("copies stored", 9, 10))))
        print("   first copy: scalar columns + geometry +%d, ring +%d, rest +%d" % (out[916] - g[9], out[917] - out[916], g[10] - out[917]))
