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
# one translation unit per (grid geometry, kernel family) + the host side: see units()
GEOMS = {"car": 0, "oct": 1, "vor": 2, "amr": 3, "sph": 4, "cyl": 5}
# -ffp-contract=off: the cell-walk arithmetic must round like the reference formulation (see find_wall)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-ffp-contract=off", "-fPIC"]
OBJDIR = os.path.join(os.path.dirname(HERE), "build", "obj")


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


HOST_UNITS = ("create", "lucy", "imaging", "engine")      # the host side of the C-ABI (hyp_engine.h); heaviest first
PARTS = {"lucy": 0, "tile": 1, "final": 2, "defer": 3, "ray": 4, "finalp": 5}
# heaviest first: the pool starts them first so that the long poles do not land at the end (seconds per unit on this container's
# cores, HYP_BUILD_TIMES=1, round 6: defer 100-139, tile 49-83, final 45-64, lucy 33, finalp 14-18, ray < 14)
_COST = {"defer": 6, "tile": 5, "final": 4, "lucy": 3, "finalp": 2, "ray": 1}


def units():
    """(name, source, defines): the host side + one unit per (geometry, kernel family) -- hyp_geom.hip's header."""
    u = [(name, "hyp_%s.hip" % name, []) for name in HOST_UNITS]
    for part in sorted(PARTS, key=lambda k: -_COST[k]):
        for g, k in GEOMS.items():
            u.append(("%s_%s" % (part, g), "hyp_geom.hip", ["-DHYP_GEOM_TU=%d" % k, "-DHYP_PART=%d" % PARTS[part]]))
    return u


def _deps_of(depfile):
    """Prerequisites listed in a make-style dependency file written by `hipcc -MD -MF`."""
    try:
        text = open(depfile).read()
    except OSError:
        return None
    text = text.replace("\\\n", " ")
    return [t for t in text.split(":", 1)[1].split() if t] if ":" in text else None


def unit_is_stale(obj, cmd, cwd=CSRC):
    """An object is rebuilt when it is missing, when its command line changed, or when a file it was compiled from is newer."""
    if not os.path.exists(obj):
        return True
    try:
        if open(obj + ".cmd").read() != " ".join(cmd):
            return True
    except OSError:
        return True
    deps = _deps_of(obj + ".d")
    if deps is None:
        return True
    t = os.path.getmtime(obj)
    for d in deps:
        path = d if os.path.isabs(d) else os.path.join(cwd, d)
        if not os.path.exists(path) or os.path.getmtime(path) > t:
            return True
    return False


def _unit_cmd(src, defs, obj, extra_flags):
    return [_hipcc()] + HIPCC_FLAGS + list(extra_flags) + defs + ["-MD", "-MF", obj + ".d", "-c", src, "-o", obj]


def is_stale(objdir=OBJDIR):
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if not os.path.isdir(objdir) or not any(f.endswith(".o") for f in os.listdir(objdir)):
        # a snapshot without the objects (build/obj does not travel to the GPU box): the library against its sources
        deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
        deps.append(os.path.join(os.path.dirname(HERE), "include", "hyperion_amd.h"))
        return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))
    for name, src, defs in units():
        obj = os.path.join(objdir, name + ".o")
        if unit_is_stale(obj, _unit_cmd(src, defs, obj, ())) or os.path.getmtime(obj) > t:
            return True
    return False


def compile_units(out, extra_flags=(), objdir=OBJDIR, verbose=False, force=False, jobs=None, select=None):
    """hipcc -c every stale translation unit, at most `jobs` at a time, then link all of them into `out`.
    `select`: only units whose name contains one of these strings are (re)compiled -- tuning builds."""
    os.makedirs(objdir, exist_ok=True)
    jobs = jobs or max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("HYP_BUILD_JOBS", "64"))))
    todo, objs = [], []
    for name, src, defs in units():
        obj = os.path.join(objdir, name + ".o")
        objs.append(obj)
        if select is not None and not any(x in name for x in select):
            continue
        cmd = _unit_cmd(src, defs, obj, extra_flags)
        if force or unit_is_stale(obj, cmd):
            todo.append((cmd, obj))
    import time
    running, failed = [], None
    while (todo or running) and failed is None:
        while todo and len(running) < jobs:
            cmd, obj = todo.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            for stale in (obj + ".cmd", obj):
                if os.path.exists(stale):
                    os.remove(stale)
            running.append((cmd, obj, subprocess.Popen(cmd, cwd=CSRC), time.time()))
        still = []
        for cmd, obj, p, t_start in running:
            rc = p.poll()
            if rc is None:
                still.append((cmd, obj, p, t_start))
            elif rc != 0:
                failed = failed or (rc, cmd)
            else:
                with open(obj + ".cmd", "w") as f:
                    f.write(" ".join(cmd))
                if os.environ.get("HYP_BUILD_TIMES"):      # which units the cold build waits for
                    print("%6.1f s  %s" % (time.time() - t_start, os.path.basename(obj)), flush=True)
        running = still
        if running and failed is None:
            time.sleep(0.2)
    for _, _, p, _ in running:
        p.wait()
    if failed:
        raise subprocess.CalledProcessError(*failed)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(out + ".tmp", out)
    return out


def build_extension(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    return compile_units(LIB, verbose=verbose, force=force)


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
