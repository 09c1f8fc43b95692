#!/opt/conda/bin/python3.9
"""Read gpurun_out/car_options.gpu.rtout (written on the GPU box by tests/test_gpu_run.py from the reference-written
tests/golden/car_options.rtin) with the REFERENCE's own ModelOutput: the copied input, the filter image group, the 4-byte
cubes and the n_photons / density_diff / specific_energy_spectrum datasets.  Build container only.
    /opt/conda/bin/python3.9 tools/validate_options_rtout_with_reference.py gpurun_out/car_options.gpu.rtout
"""
import os, sys, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.environ.get("HYPERION_REFERENCE_COPY", "/tmp/hyp_probe"))
import numpy as np
for name, fn in [("asscalar", lambda a: a.item()), ("alen", lambda a: len(a))]:
    if not hasattr(np, name): setattr(np, name, fn)
for name, t in [("float", float), ("int", int), ("bool", bool), ("object", object), ("str", str), ("complex", complex)]:
    if not hasattr(np, name): setattr(np, name, t)
from hyperion.model import ModelOutput

m = ModelOutput(sys.argv[1])          # copy_input = yes: no external link to repair
im = m.get_image(group=0, inclination=0, distance=None, units="ergs/s")
print("filter group image", im.val.shape, "nu", im.nu, "wav", im.wav, "total %.4e" % np.nansum(im.val))
assert im.val.shape[-1] == 1 and abs(im.wav[0] - 2.0) < 1e-6
s = m.get_sed(group=1, inclination="all", aperture=-1, distance=None, units="ergs/s")
print("plain group SED", s.val.shape, "total %.4e" % np.nansum(s.val))
for it in (0, 1):
    q = m.get_quantities(iteration=it)
    print("iteration", it, "quantities", sorted(q.quantities.keys()))
q = m.get_quantities()
assert "n_photons" in q.quantities and "density_diff" in q.quantities and "specific_energy_spectrum" in q.quantities, sorted(q.quantities)
print("n_photons max", int(q["n_photons"].array.max()), "spectrum shape", np.shape(q["specific_energy_spectrum"][0].array) if hasattr(q["specific_energy_spectrum"], "__getitem__") else None)
print("OK: the reference's ModelOutput reads every optional output")
