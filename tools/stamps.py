"""Diagnostic: builds a -DSZ_STAMPS copy of the library and prints the stamp timeline of one lane
group (block 0, group 0) of the narrow-phase kernel: stage id and cycles since the wave started."""
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
names = {0: "rings staged", 1: "candidate edges done", 2: "detect signs done", 3: "detect params done", 4: "canonical done",
         5: "containment done", 6: "rank done", 7: "trace+area done", 8: "match/many done", 10: "intersects done",
         11: "item done", 12: "overlap tests done", 13: "direction settled", 14: "certified check done (detect-only clip + exact area change)",
         20: "stop flag read", 21: "segment + housekeeping", 22: "work item read", 23: "ring offsets read", 24: "ring loads back", 25: "scalar loads back", 26: "sign + box loads back"}
for blk in range(0, 40):
    os.environ["SZ_DEBUG"] = str((blk << 8) | int(os.environ.get("SZ_STAMPS_TWICE", "0")) * 16)
    w = fields.build_world(subzero_jl_amd.World(0), cfg)
    w.run(3, 0, cfg["dt"], coupling_dt=1)
    out = np.zeros(512 + 8 * 8000, np.int64)
    w._chk(w.L.sz_debug_stamps(w.h, capi.ptr(out, capi._lp)))
    w.run(1, 3, cfg["dt"], coupling_dt=1)
    w._chk(w.L.sz_debug_stamps(w.h, capi.ptr(out, capi._lp)))
    if out[0] > 0:
        print("block", blk, "group 0: timeline of a pair with contact rows (cycles)")
        prev = 0
        for e in out[1:1 + int(out[0])]:
            k, t = int(e) >> 48, int(e) & ((1 << 48) - 1)
            print(f"  {names.get(k, k):24s} t={t:8d}  +{t - prev:7d}")
            prev = t
        break

# lifetimes of all wavefronts of the narrow kernel in the last step (4096-cycle buckets)
w = fields.build_world(subzero_jl_amd.World(0), cfg)
os.environ["SZ_DEBUG"] = str(1 << 30)
w.run(3, 0, cfg["dt"], coupling_dt=1)
out = np.zeros(512 + 8 * 8000, np.int64)
w._chk(w.L.sz_debug_stamps(w.h, capi.ptr(out, capi._lp)))
w.run(1, 3, cfg["dt"], coupling_dt=1)
w._chk(w.L.sz_debug_stamps(w.h, capi.ptr(out, capi._lp)))
h = out[256:384]
tot = h.sum()
print("wavefront lifetimes (k cycles: count):", {int(4 * k): int(v) for k, v in enumerate(h) if v})
cum = np.cumsum(h) / max(tot, 1)
print("median %.0f k cycles, 90 %% %.0f k, 99 %% %.0f k, max %.0f k" % tuple(4.096 * (np.searchsorted(cum, q) + 1) for q in (0.5, 0.9, 0.99, 1.0)))
for r in range(5):
    if out[401 + 2 * r]:
        print(f"wavefronts whose heaviest item made {r} rows: {int(out[401 + 2 * r])}, mean lifetime {out[400 + 2 * r] / out[401 + 2 * r] / 1e3:.0f} k cycles")
