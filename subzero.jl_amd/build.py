"""Builds libsubzero_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "sz_api.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in sorted(f for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith(".hpp"))] + \
       [os.path.join(os.path.dirname(HERE), "include", "subzero_hip.h")]
LIB = os.path.join(HERE, "libsubzero_hip.so")
# -ffp-contract=off: fp64 expressions evaluate as written (no FMA), which is what makes the
# discrete decisions of the narrow phase agree with the CPU reference path.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]
# experiments (tools/probe/*): extra -D switches for an A/B build, e.g. SZ_EXTRA_FLAGS="-DFRC_PLAIN_LANES=16"
FLAGS += os.environ.get("SZ_EXTRA_FLAGS", "").split()


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build(force=False):
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in DEPS):
        return LIB
    subprocess.check_call([hipcc()] + FLAGS + ["-o", LIB, SRC])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
