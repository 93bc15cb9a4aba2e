"""durations of one kernel, launch by launch, from the last tools/kstats_n.sh trace: python tools/ktrace_series.py sz_k_integrate"""
import csv, glob, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "sz_k_integrate"
f = glob.glob("gpurun_out/prof_ksn/**/*kernel_trace.csv", recursive=True)[0]
d = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
d.sort()
print(" ".join(f"{x[1]:.0f}" for x in d))
