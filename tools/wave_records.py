"""Diagnostic (-DSZ_STAMPS build): one record per wavefront of the narrow kernel in one step -- lifetime, passes of the
check loop, check tasks, live items, cycles in phase A / B / C -- grouped so that the tail of the launch can be read."""
import os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from subzero_jl_amd import build as b
lib = os.path.join(ROOT, "subzero.jl_amd", "libsubzero_hip_stamps.so")
subprocess.check_call([b.hipcc()] + b.FLAGS + ["-DSZ_STAMPS", "-o", lib, b.SRC])
b.LIB = lib
import subzero_jl_amd
from subzero_jl_amd import fields, capi
capi._LIB = None
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfg = fields.make_config(n_floes=n, seed=12345)
os.environ["SZ_DEBUG"] = str(1 << 30)
w = fields.build_world(subzero_jl_amd.World(0), cfg)
w.run(20, 0, cfg["dt"], coupling_dt=1)
N = 512 + 8 * 8000
out = np.zeros(N, np.int64)
w._chk(w.L.sz_debug_stamps(w.h, capi.ptr(out, capi._lp)))
w.run(1, 20, cfg["dt"], coupling_dt=1)
w._chk(w.L.sz_debug_stamps(w.h, capi.ptr(out, capi._lp)))
nrec = int(min(out[511], 8000))
r = out[512:512 + 8 * nrec].reshape(nrec, 8)
r = r[(r[:, 4] + r[:, 5] > 0) & (r[:, 1] > 0) & (r[:, 4] < (1 << 40))]      # wavefronts that ran a round (a record of a wavefront without items holds no phase times)
a1 = ((r[:, 7] >> 8) & ((1 << 28) - 1)) << 8; a2 = (r[:, 7] >> 36) << 8; r[:, 7] &= 255
pro = (r[:, 1] >> 8) << 8; r[:, 1] &= 255          # cycles from the wavefront's start to its first item (prologue)
print("phase A split, mean k cycles: staging %.1f  contact clip %.1f  after the clip (overlap tests, vertex matching, directions) %.1f" % (a1.mean() / 1e3, a2.mean() / 1e3, (r[:, 4] - a1 - a2).mean() / 1e3))
heavy = r[:, 7] >= 2
print("   ... of the wavefronts with a 2-row item: staging %.1f  clip %.1f  after %.1f ; the others: %.1f  %.1f  %.1f" % (a1[heavy].mean() / 1e3, a2[heavy].mean() / 1e3, (r[heavy, 4] - a1[heavy] - a2[heavy]).mean() / 1e3, a1[~heavy].mean() / 1e3, a2[~heavy].mean() / 1e3, (r[~heavy, 4] - a1[~heavy] - a2[~heavy]).mean() / 1e3))
print(f"{len(r)} wavefronts with work of {nrec}; lifetime k cycles: median {np.median(r[:,0])/1e3:.0f}, 90% {np.percentile(r[:,0],90)/1e3:.0f}, 99% {np.percentile(r[:,0],99)/1e3:.0f}, max {r[:,0].max()/1e3:.0f}")
print("mean cycles (k): phase A %.1f  B %.1f  C %.1f" % tuple(r[:, 4:7].mean(0) / 1e3))
print("by passes of the check loop:")
for p in sorted(set(r[:, 1])):
    q = r[r[:, 1] == p]
    print(f"  passes {p}: {len(q):5d} wavefronts, lifetime mean {q[:,0].mean()/1e3:6.1f} k  max {q[:,0].max()/1e3:6.1f} k   A {q[:,4].mean()/1e3:5.1f}  B {q[:,5].mean()/1e3:5.1f}  C {q[:,6].mean()/1e3:5.1f}  tasks {q[:,2].mean():.1f}  live {q[:,3].mean():.1f}")
print("the 12 longest:")
for q in r[np.argsort(-r[:, 0])[:12]]:
    print(f"  lifetime {q[0]/1e3:6.1f} k  passes {q[1]}  tasks {q[2]}  live {q[3]}  A {q[4]/1e3:5.1f}  B {q[5]/1e3:5.1f}  C {q[6]/1e3:5.1f}  max rows {q[7]}")
print("phase A by live items:")
for l in sorted(set(r[:, 3])):
    q = r[r[:, 3] == l]
    print(f"  live {l}: {len(q):5d} wavefronts  A mean {q[:,4].mean()/1e3:5.1f} k  max {q[:,4].max()/1e3:5.1f} k")
# round 4: where the slowest wavefronts spend their time -- the launch lasts as long as its slowest wavefront
order = np.argsort(-r[:, 0])
rest = r[:, 4] - a1 - a2
def row(name, sel):
    q = r[sel]
    print(f"  {name:28s} n={len(q):5d}  lifetime {q[:,0].mean()/1e3:6.1f} k | prologue {pro[sel].mean()/1e3:5.1f}  epilogue {(q[:,0]-pro[sel]-q[:,4]-q[:,5]-q[:,6]).mean()/1e3:5.1f} | staging {a1[sel].mean()/1e3:5.1f}  contact clip {a2[sel].mean()/1e3:5.1f}  after the clip {rest[sel].mean()/1e3:5.1f}  checks (B) {q[:,5].mean()/1e3:5.1f}  rows (C) {q[:,6].mean()/1e3:4.1f}  | passes {q[:,1].mean():.2f} tasks {q[:,2].mean():.1f} live {q[:,3].mean():.1f}")
print("phase split by lifetime rank (mean k cycles):")
n = len(r)
row("slowest 1 %", order[:max(1, n // 100)])
row("slowest 5 %", order[:max(1, n // 20)])
row("slowest 25 %", order[:max(1, n // 4)])
row("middle half", order[n // 4:3 * n // 4])
row("fastest 25 %", order[3 * n // 4:])
print("percentiles (k cycles)   50%    90%    99%    max")
for name, v in (("staging", a1), ("contact clip", a2), ("after the clip", rest), ("checks (B)", r[:, 5]), ("lifetime", r[:, 0])):
    print(f"  {name:20s} {np.percentile(v,50)/1e3:6.1f} {np.percentile(v,90)/1e3:6.1f} {np.percentile(v,99)/1e3:6.1f} {v.max()/1e3:6.1f}")
