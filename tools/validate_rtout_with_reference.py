#!/opt/conda/bin/python3.9
"""Read a .rtout written by `python -m hyperion_amd` with the REFERENCE's own
ModelOutput (needs /root/reference staged as in tests/golden/make_fixtures.py;
runs only in the build container).  usage:
    /opt/conda/bin/python3.9 tools/validate_rtout_with_reference.py gpurun_out/car_peeloff.False.gpu.rtout tests/golden/car_peeloff.False.rtin
"""
import os, shutil, sys, tempfile, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.environ.get("HYPERION_REFERENCE_COPY", "/tmp/hyp_probe"))
import numpy as np
for name, fn in [("asscalar", lambda a: a.item()), ("alen", lambda a: len(a))]:
    if not hasattr(np, name): setattr(np, name, fn)
for name, t in [("float", float), ("int", int), ("bool", bool), ("object", object), ("str", str), ("complex", complex)]:
    if not hasattr(np, name): setattr(np, name, t)
import h5py
from hyperion.model import ModelOutput

out, inp = sys.argv[1], os.path.abspath(sys.argv[2])
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "model.rtout")
shutil.copy(out, path)
with h5py.File(path, "r+") as f:          # the Input link points to the GPU box's path
    del f["Input"]
    f["Input"] = h5py.ExternalLink(inp, "/")
m = ModelOutput(path)
for g in range(3):
    s = m.get_sed(group=g, inclination="all", aperture=-1, distance=None, units="ergs/s")
    print("group", g, "SED", s.val.shape, "total %.4e" % np.nansum(s.val))
    im = m.get_image(group=g, inclination=0, distance=None, units="ergs/s")
    print("group", g, "image", im.val.shape, "total %.4e" % np.nansum(im.val))
q = m.get_quantities()
print("quantities", sorted(q.quantities.keys()), "mean T = %.3f K" % q["temperature"][0].array.mean())
print("OK: the reference's ModelOutput reads the file")
