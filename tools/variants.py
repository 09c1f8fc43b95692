"""Build tuning variants of the HIP library into build/variants/ (they travel to
the GPU box with the snapshot) and, on the GPU box, time each one.
  python tools/variants.py build name1:"-DFOO=1 -DBAR=2" name2:"..."
  python tools/variants.py run [photons] [option=value ...]
"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "build", "variants")


def build(specs):
    """Each variant: hyp_engine.hip + the units of ONE geometry (env GEOM, default Cartesian).  The kernel families named by env
    PARTS (default "tile") are compiled with the variant's flags for ONE species count (env ND, default 1); the others are taken
    from the default build (build/obj), so a variant costs two compilations.  At most 8 compilers run at a time."""
    from hyperion_amd.build import CSRC, HIPCC_FLAGS, _hipcc, PARTS, GEOMS, OBJDIR
    os.makedirs(VDIR, exist_ok=True)
    geom = int(os.environ.get("GEOM", "0"))      # GEOM_* of the one geometry linked (0 Cartesian, 1 octree, ...)
    gname = [g for g, k in GEOMS.items() if k == geom][0]
    nd = os.environ.get("ND", "1")
    mine = os.environ.get("PARTS", "tile").split(",")
    jobs, links = [], []
    for spec in specs:
        name, _, flags = spec.partition(":")
        out = os.path.join(VDIR, name + ".so")
        objs = []
        from hyperion_amd.build import HOST_UNITS
        units = [("host_" + u, "hyp_%s.hip" % u, ["-DHYP_VARIANT_GEOM=%d" % geom]) for u in HOST_UNITS]
        for part, k in PARTS.items():
            if part in mine:
                units.append((part, "hyp_geom.hip", ["-DHYP_GEOM_TU=%d" % geom, "-DHYP_PART=%d" % k, "-DHYP_ONLY_ND=" + nd]))
            else:
                objs.append(os.path.join(OBJDIR, "%s_%s.o" % (part, gname)))
        for unit, src, defs in units:
            obj = os.path.join(VDIR, "%s_%s.o" % (name, unit))
            jobs.append((name, [_hipcc()] + HIPCC_FLAGS + flags.split() + defs + ["-c", src, "-o", obj, "-Rpass-analysis=kernel-resource-usage"]))
            objs.append(obj)
        links.append((name, [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", out]))
    running = []

    def reap(p_name, p):
        err = p.communicate()[1]
        for b in err.split("Function Name: "):
            if b.startswith(os.environ.get("KERNEL", "_Z11lucy_kernelILi1E")):
                info = [l.split("remark:")[1].strip() for l in b.split("\n") if "remark:" in l and
                        any(k in l for k in (" VGPRs:", "AGPRs:", "ScratchSize", "Occupancy", "VGPRs Spill"))]
                print(p_name, "|", "; ".join(info), flush=True)
        if p.returncode:
            print(p_name, "rc", p.returncode, err[-2000:])
    while jobs or running:
        while jobs and len(running) < 8:
            name, cmd = jobs.pop(0)
            running.append((name, subprocess.Popen(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)))
        name, p = running.pop(0)
        reap(name, p)
    for name, cmd in links:
        print(name, "link rc", subprocess.call(cmd, cwd=CSRC))


def run_one(lib, photons, opts):
    import hyperion_amd.engine as E
    import ctypes as C
    E._lib = None
    L = E.load_library(lib)
    E._lib = L
    import hyperion_amd
    from hyperion_amd.benchmark import make_benchmark_problem
    p = make_benchmark_problem(128)
    eng = hyperion_amd.Engine(p)
    for k, v in opts.items():
        eng.set_option(k, v)
    eng.lucy_iteration(photons // 4, 1, want_output=False)
    best = 1e30
    for it in (2, 3):
        _, st = eng.lucy_iteration(photons, it, want_output=False)
        best = min(best, eng.last_kernel_ms()[0])
    print(json.dumps({"lib": os.path.basename(lib), "opts": opts, "ms": best, "packets_per_s": photons / best * 1e3,
                      "crossings_per_s": st["crossings"] / best * 1e3}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "one":
        run_one(sys.argv[2], int(float(sys.argv[3])), {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[4:]})
    else:
        photons = sys.argv[2] if len(sys.argv) > 2 else "2e7"
        for f in sorted(os.listdir(VDIR)):
            if f.endswith(".so"):
                subprocess.call([sys.executable, __file__, "one", os.path.join(VDIR, f), photons] + sys.argv[3:])
