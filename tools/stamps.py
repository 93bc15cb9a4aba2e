"""Diagnostic: builds a -DSZ_STAMPS copy of the library and prints cycles per narrow-phase stage."""
import ctypes, os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from subzero_jl_amd import build as b
lib = os.path.join(ROOT, "subzero.jl_amd", "libsubzero_hip_stamps.so")
subprocess.check_call([b.hipcc()] + b.FLAGS + ["-DSZ_STAMPS", "-o", lib, b.SRC])
b.LIB = lib
import subzero_jl_amd
from subzero_jl_amd import fields, capi
capi._LIB = None
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
cfg = fields.make_config(n_floes=n, seed=12345)
w = fields.build_world(subzero_jl_amd.World(0), cfg)
w.run(3, 0, cfg["dt"], coupling_dt=1)
out = np.zeros(16, np.int64)
w._chk(w.L.sz_debug_stamps(w.h, capi.ptr(out, capi._lp)))
w.profile(True)
w.run(10, 3, cfg["dt"], coupling_dt=1)
w._chk(w.L.sz_debug_stamps(w.h, capi.ptr(out, capi._lp)))
P = w.stats()["n_pairs"]
names = ["load rings", "bbox", "detect signs", "detect params", "canonical", "(K==0 pre)", "containment", "rank", "trace+area",
         "match/many", "rows pre-check", "dir check", "rows tail+store"]
names = ["0 load", "1 bbox", "2 detect-signs", "3 detect-params", "4 canonical", "5 containment", "6 rank", "7 trace+area",
         "8 match/many", "9 rows-pre", "10 dircheck-intersects", "11 tail"]
print("pairs", P, "steps 10; cycles per pair per stage (100 MHz ticks? see clock64):")
for k, nm in enumerate(names):
    print(f"  {nm:24s} {out[k] / (10 * P):10.1f}")
print("  total", out[:12].sum() / (10 * P))
print("  groups", out[13] / 10, "groups with items", out[14] / 10, "mean wave lifetime ticks", out[12] / max(out[13], 1))
kt = w.kernel_times(); print("  narrow kernel ms", kt["narrow"][0] / max(kt["narrow"][1], 1))
