#!/usr/bin/env python
"""Writes tests/golden/killed_counts.json: the killed-packet counters of EVERY golden .rtout the reference ships
(hyperion/model/tests/data/*.rtout) -- root attributes killed_photons_{geo,int}_{final,raytracing} and the per-iteration
killed_photons_{geo,int} (src/main/main.f90:241-246,287-288,316-317).  Data only; run in the build container with the
interpreter that has h5py:

    /opt/conda/bin/python3.9 tests/golden/make_killed_fixture.py /root/reference
"""
import glob
import json
import os
import sys

import h5py

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out = {}
for path in sorted(glob.glob(os.path.join(ref, "hyperion", "model", "tests", "data", "*.rtout"))):
    with h5py.File(path, "r") as f:
        its = sorted(k for k in f if k.startswith("iteration_"))
        ent = {"iterations": [[int(f[k].attrs["killed_photons_geo"]), int(f[k].attrs["killed_photons_int"])] for k in its]}
        for part in ("final", "raytracing"):
            if "killed_photons_geo_" + part in f.attrs:
                ent[part] = [int(f.attrs["killed_photons_geo_" + part]), int(f.attrs["killed_photons_int_" + part])]
        out[os.path.basename(path)[:-len(".rtout")]] = ent
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "killed_counts.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
print("%d goldens -> %s" % (len(out), dst))
