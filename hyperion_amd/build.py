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
SOURCES = ["hyp_engine.hip", "hyp_geom.hip", "hyp_kernels.h", "hyp_device.h", "hyp_tiled.h", "hyp_pick.h", "hyp_polar.h", "hyp_epilogue.h", "hyp_defer.h", "hyp_vtile.h", "hyp_otile.h", "hyp_atile.h", "hyp_stage.h"]
# one translation unit per grid geometry (lucy / final / ray kernels x species counts) + the host side
GEOMS = {"car": 0, "oct": 1, "vor": 2, "amr": 3, "sph": 4, "cyl": 5}
# -ffp-contract=off: the cell-walk arithmetic must round like the reference formulation (see find_wall)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-ffp-contract=off", "-fPIC"]
OBJDIR = os.path.join(os.path.dirname(HERE), "build", "obj")


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


def compile_units(out, extra_flags=(), objdir=OBJDIR, verbose=False):
    """hipcc -c every translation unit in parallel, then link them into `out`."""
    os.makedirs(objdir, exist_ok=True)
    units = [("engine", "hyp_engine.hip", [])] + [("geom_" + g, "hyp_geom.hip", ["-DHYP_GEOM_TU=%d" % k]) for g, k in GEOMS.items()]
    procs = []
    for name, src, defs in units:
        obj = os.path.join(objdir, name + ".o")
        cmd = [_hipcc()] + HIPCC_FLAGS + list(extra_flags) + defs + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, obj, subprocess.Popen(cmd, cwd=CSRC)))
    objs = []
    for cmd, obj, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
        objs.append(obj)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(out + ".tmp", out)
    return out


def build_extension(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    return compile_units(LIB, verbose=verbose)


# ---- the native .rtin -> .rtout driver (hyp_run.cpp): `hyperion_car`-style executables over the C ABI -------------------
BIN = os.path.join(HERE, "bin")
DRIVER = os.path.join(BIN, "hyperion_amd_run")
GRID_SUFFIXES = ("car", "sph", "cyl", "oct", "amr", "vor")      # scripts/hyperion:44-58 calls hyperion_<suffix>


def hdf5_prefix():
    """Where a C libhdf5 lives (headers + shared library), or None: this image ships one with /opt/conda."""
    for prefix in (os.environ.get("HDF5_DIR"), "/opt/conda", "/usr", "/usr/local"):
        if prefix and os.path.exists(os.path.join(prefix, "include", "hdf5.h")) and os.path.exists(os.path.join(prefix, "lib", "libhdf5.so")):
            return prefix
    return None


def build_native_driver(force=False, verbose=False):
    """g++ hyp_run.cpp against libhdf5 and libhyperion_amd.so -> hyperion_amd/bin/hyperion_amd_run plus the names the
    reference's launcher looks for (hyperion_car, hyperion_sph, ...).  Returns the path, or None where there is no libhdf5
    to link (the Python adapter `python -m hyperion_amd` is the file seam then)."""
    prefix = hdf5_prefix()
    if prefix is None:
        return None
    src = os.path.join(CSRC, "hyp_run.cpp")
    hdr = os.path.join(os.path.dirname(HERE), "include", "hyperion_amd.h")
    if not force and os.path.exists(DRIVER) and os.path.getmtime(DRIVER) > max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(LIB)):
        return DRIVER
    os.makedirs(BIN, exist_ok=True)
    # libstdc++ is linked statically and the system directories are searched first at link time: the prefix that has
    # libhdf5 (conda) may carry an older libstdc++ than the HIP runtime needs; at run time RUNPATH only serves the
    # executable's own two dependencies
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-I" + os.path.join(prefix, "include"), "-isystem", os.path.join(rocm, "include"), src, "-o", DRIVER + ".tmp",
           "-L" + CSRC, "-lhyperion_amd", os.path.join(prefix, "lib", "libhdf5.so"),
           "-L" + os.path.join(rocm, "lib"), "-lrccl", "-lamdhip64", "-Wl,-rpath," + os.path.join(rocm, "lib"), "-static-libstdc++", "-static-libgcc",
           "-Wl,-rpath-link,/usr/lib/x86_64-linux-gnu", "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--enable-new-dtags",
           "-Wl,-rpath,$ORIGIN/../csrc", "-Wl,-rpath," + os.path.join(prefix, "lib")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(DRIVER + ".tmp", DRIVER)
    # scripts/hyperion:44-92 starts hyperion_<grid> or, with -m N, `mpirun -n N hyperion_<grid>_mpi`: the same executable
    # (it is a rank of N when the launcher's environment says so)
    for suffix in GRID_SUFFIXES:
        for name in ("hyperion_" + suffix, "hyperion_" + suffix + "_mpi"):
            link = os.path.join(BIN, name)
            if os.path.lexists(link):
                os.remove(link)
            os.symlink("hyperion_amd_run", link)
    return DRIVER


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))
    print(build_native_driver(force=True, verbose=True))
