"""Builds the HIP extension ``hyperion_amd/csrc/libhyperion_amd.so`` for gfx950
with hipcc (cross-compiles without a GPU).  In-tree so the built library travels
with the source snapshot."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libhyperion_amd.so")
SOURCES = ["hyp_engine.hip", "hyp_kernels.h", "hyp_device.h", "hyp_tiled.h"]
# -ffp-contract=off: the cell-walk arithmetic must round like the reference formulation (see find_wall)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-ffp-contract=off", "-fPIC", "-shared"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "hyperion_amd.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_extension(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc()] + HIPCC_FLAGS + ["hyp_engine.hip", "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))
