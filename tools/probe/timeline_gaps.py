"""gaps between consecutive kernels on the device, from the kernel trace of the last tools/kstats_args.sh run: per kernel name the mean time
from the END of the launch before it to its START (what the launch boundary + whatever the host did in between cost), over the timed steps"""
import csv, glob, sys, collections
R = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_ksa"
f = glob.glob(R + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]          # the second half: the timed blocks
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    n = b["Kernel_Name"].split("(")[0][-40:]
    gap[n].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
    dur[n].append(int(b["End_Timestamp"]) - int(b["Start_Timestamp"]))
tot = 0
for n in sorted(gap, key=lambda k: -len(gap[k]))[:12]:
    g = sorted(gap[n]); d = sorted(dur[n])
    print(f"{n:42s} n={len(g):6d}  gap before: median {g[len(g)//2]/1e3:6.2f} us mean {sum(g)/len(g)/1e3:6.2f}   duration: median {d[len(d)//2]/1e3:6.2f} mean {sum(d)/len(d)/1e3:6.2f}")
