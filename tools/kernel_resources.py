#!/usr/bin/env python3
"""Registers / LDS / scratch of every kernel in the built library, read from the code object's metadata (no GPU needed).

    python tools/kernel_resources.py [substring ...]
"""
import os, re, subprocess, sys, tempfile

R = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = "/opt/rocm/lib/llvm/bin"
lib = os.path.join(R, "subzero.jl_amd", "libsubzero_hip.so")


def resources():
    """{demangled kernel name without arguments: {"vgpr", "agpr", "sgpr", "lds", "scratch"}} of the built library"""
    with tempfile.TemporaryDirectory() as t:
        fat, co = os.path.join(t, "fat.bin"), os.path.join(t, "dev.co")
        subprocess.check_call([f"{L}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.devnull])
        subprocess.check_call([f"{L}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}", "--unbundle"])
        notes = subprocess.check_output([f"{L}/llvm-readelf", "--notes", co], text=True)
    rows, row = [], {}
    for line in notes.splitlines():
        m = re.match(r"^  (- | {2})\.(\w+):\s+(.*)$", line)           # kernel-level keys only (argument keys are indented deeper)
        if not m:
            continue
        if m.group(1) == "- " and row:
            rows.append(row); row = {}
        row[m.group(2)] = m.group(3).strip().strip("'")
    if row:
        rows.append(row)
    names = [r.get("name", "?") for r in rows]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    out = {}
    for r, d in zip(rows, dem):
        d = re.sub(r"\(.*", "", d).replace("void ", "").replace("sz::", "")
        out[d] = {"vgpr": int(r.get("vgpr_count", -1)), "agpr": int(r.get("agpr_count", 0)), "sgpr": int(r.get("sgpr_count", -1)),
                  "lds": int(r.get("group_segment_fixed_size", -1)), "scratch": int(r.get("private_segment_fixed_size", -1))}
    return out


if __name__ == "__main__":
    want = sys.argv[1:]
    for d, r in sorted(resources().items()):
        if want and not any(w in d for w in want):
            continue
        print(f"{d[:64]:66s} vgpr {r['vgpr']:4d} agpr {r['agpr']:3d} sgpr {r['sgpr']:4d} lds {r['lds']:6d} scratch {r['scratch']:5d}")
