"""Per-kernel averages of the rocprofv3 passes made by tools/profile_round.sh.

    python3 tools/pmc_summary.py <dir with prof_fetch/ prof_write/ prof_kt/> <tag> [workload:n_floes]

Writes <dir>/<tag>_pmc_fetch_size.csv, <tag>_pmc_write_size.csv, <tag>_bench_kernel_stats.csv and merges the HBM bytes
per launch of every kernel (gfx950: 2 x FETCH_SIZE + WRITE_SIZE, both counters in KB, MI355X_MICROARCH.md) into
<dir>/pmc_traffic.json under the key `workload:n_floes` -- the table bench.py reads from profiles/ for
roofline.traffic (copy it there together with the csv files it was made from)."""
import csv, glob, json, os, sys
from collections import defaultdict
root, tag = sys.argv[1], sys.argv[2]
key = sys.argv[3] if len(sys.argv) > 3 else "configs1:10000"


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def norm(name):          # "sz::sz_k_narrow<8, 20, 8, 16, 4, 64, 0, 0, 3>" -> "sz_k_narrow<8,20,8,16,4,64,0,0,3>"
    return short(name).replace("sz::", "").replace(" ", "")


kb = {}
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
            kb.setdefault(norm(k), {})[counter] = v / n
    print(out)
if kb:
    path = os.path.join(root, "pmc_traffic.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    table[key] = {k: {"fetch_size_kb": v.get("FETCH_SIZE"), "write_size_kb": v.get("WRITE_SIZE"),
                      "hbm_bytes_per_launch": (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0,
                      "source": f"profiles/{tag}_pmc_fetch_size.csv, profiles/{tag}_pmc_write_size.csv (2 x FETCH_SIZE + WRITE_SIZE)"}
                  for k, v in kb.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
    # the sources the passes were made on: bench.py only quotes the table for the same kernel code
    import hashlib
    h = hashlib.sha256()
    cs = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "subzero.jl_amd", "csrc")
    for f in sorted(os.listdir(cs)):
        if f.endswith((".hpp", ".hip")):
            h.update(open(os.path.join(cs, f), "rb").read())
    table[key]["_meta"] = {"kernel_source_sha16": h.hexdigest()[:16], "tag": tag}
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
    print(path)
f = glob.glob(os.path.join(root, "prof_kt", "**", "*kernel_stats.csv"), recursive=True)
if f:
    out = os.path.join(root, f"{tag}_bench_kernel_stats.csv")
    open(out, "w").write(open(f[0]).read())
    print(out)
