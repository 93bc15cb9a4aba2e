"""Per-kernel averages of the rocprofv3 passes made by tools/profile_round.sh."""
import csv, glob, os, sys
from collections import defaultdict
root, tag = sys.argv[1], sys.argv[2]


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


for sub, counter in (("prof_fetch", "FETCH_SIZE"), ("prof_write", "WRITE_SIZE")):
    f = glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        a = acc[short(r["Kernel_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    out = os.path.join(root, f"{tag}_pmc_{counter.lower()}.csv")
    with open(out, "w") as o:
        o.write(f"kernel,launches,avg_{counter}_KB_per_launch\n")
        for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            o.write(f"\"{k}\",{n},{v / n:.3f}\n")
    print(out)
f = glob.glob(os.path.join(root, "prof_kt", "**", "*kernel_stats.csv"), recursive=True)
if f:
    out = os.path.join(root, f"{tag}_bench_kernel_stats.csv")
    open(out, "w").write(open(f[0]).read())
    print(out)
