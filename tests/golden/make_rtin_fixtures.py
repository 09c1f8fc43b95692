#!/opt/conda/bin/python3.9
"""Small self-contained .rtin inputs written by the REFERENCE front-end (hyperion.model.Model.write), one per grid
geometry, covering the source types and run modes the .rtin contract has.  They are the inputs of
tests/test_gpu_native_driver.py, which runs the native driver (hyperion_amd/bin/hyperion_<grid>) and the Python adapter on
each and compares the two .rtout files, and of tests/test_native_driver_cpu.py (input parsing without a GPU).

Run like make_fixtures.py (staged reference copy in /tmp/hyp_probe, see there):

    LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6 /opt/conda/bin/python3.9 tests/golden/make_rtin_fixtures.py
"""
import os
import sys
import tempfile
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.environ.get("HYPERION_REFERENCE_COPY", "/tmp/hyp_probe"))
warnings.filterwarnings("ignore")
import numpy as np

for name, fn in [("asscalar", lambda a: a.item()), ("alen", lambda a: len(a))]:
    if not hasattr(np, name):
        setattr(np, name, fn)
for name, t in [("float", float), ("int", int), ("bool", bool), ("object", object), ("str", str), ("complex", complex)]:
    if not hasattr(np, name):
        setattr(np, name, t)

import h5py  # noqa: E402
from hyperion.model import Model  # noqa: E402
from hyperion.grid import AMRGrid, CylindricalPolarGrid, OctreeGrid, SphericalPolarGrid, VoronoiGrid  # noqa: E402
from hyperion.dust import IsotropicDust  # noqa: E402
from hyperion.util.constants import pc, lsun, rsun  # noqa: E402


_DUST = {}


def test_dust():
    """the grey LTE test dust of hyperion/model/tests/test_helpers.py:14-18, written to a file so that Model.write can link it"""
    if "path" not in _DUST:
        d = IsotropicDust([3.e9, 3.e16], [0.5, 0.5], [1., 1.])
        d.set_lte_emissivities(10, 0.1, 1600.)
        _DUST["dir"] = tempfile.mkdtemp()
        _DUST["path"] = os.path.join(_DUST["dir"], "test_dust.hdf5")
        d.write(_DUST["path"])
    return _DUST["path"]


def keep(m, name, tmp):
    path = os.path.join(tmp, name)
    m.set_copy_input(False)
    m.write(path, copy=False, absolute_paths=True)
    out = os.path.join(HERE, name)
    with h5py.File(path, "r") as fi, h5py.File(out, "w") as fo:
        for k, v in fi.attrs.items():
            fo.attrs[k] = v
        for k in fi:
            fi.copy(k, fo, expand_external=True, expand_soft=True)
    print("wrote", out, os.path.getsize(out))


def image(m, views=((45., 45.),), n=6, **kw):
    i = m.add_peeled_images(sed=True, image=True)
    i.set_viewing_angles([v[0] for v in views], [v[1] for v in views])
    i.set_image_size(n, n)
    i.set_image_limits(-1.5 * pc, 1.5 * pc, -1.5 * pc, 1.5 * pc)
    i.set_aperture_range(2, 0.5 * pc, 2. * pc)
    i.set_wavelength_range(4, 0.1, 1000.)
    for k, v in kw.items():
        getattr(i, "set_" + k)(*v if isinstance(v, tuple) else (v,))
    return i


def octree(tmp):
    """octree; point source + limb-darkened sphere with a spot; two views, uncertainties, detailed origin tracking"""
    np.random.seed(1)
    refined = [1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0]
    m = Model()
    m.set_grid(OctreeGrid(0., 0., 0., pc, pc, pc, np.array(refined).astype(bool)))
    m.add_density_grid(np.random.random(25) * 2e-20, test_dust())
    s = m.add_point_source()
    s.luminosity, s.temperature, s.position = lsun, 5000., (0.3 * pc, -0.2 * pc, 0.1 * pc)
    s = m.add_spherical_source()
    s.luminosity, s.temperature, s.radius, s.position, s.limb = 2 * lsun, 6000., 5e4 * rsun, (-0.4 * pc, 0.1 * pc, 0.2 * pc), True
    q = s.add_spot()
    q.longitude, q.latitude, q.radius, q.luminosity, q.temperature = 40., 20., 25., 0.5 * lsun, 9000.
    image(m, views=((45., 45.), (120., 200.)), uncertainties=True, track_origin="detailed", stokes=True)
    m.set_n_initial_iterations(2)
    m.set_n_photons(initial=4000, imaging=4000)
    m.set_seed(-101)
    keep(m, "native_oct.rtin", tmp)


def amr(tmp):
    """AMR, two levels, two dust species; point collection + external box source; convergence test on"""
    np.random.seed(2)
    g = AMRGrid()
    g1 = g.add_level().add_grid()
    g1.xmin, g1.xmax, g1.ymin, g1.ymax, g1.zmin, g1.zmax = -pc, pc, -pc, pc, -pc, pc
    g1.nx, g1.ny, g1.nz = 8, 6, 4
    g1.quantities["density"] = np.random.random((4, 6, 8)) * 1e-20
    g1.quantities["density_2"] = np.random.random((4, 6, 8)) * 1e-20
    g2 = g.add_level().add_grid()
    g2.xmin, g2.xmax, g2.ymin, g2.ymax, g2.zmin, g2.zmax = -pc, 0., -pc, 0., -pc, 0.
    g2.nx, g2.ny, g2.nz = 8, 6, 4
    g2.quantities["density"] = np.random.random((4, 6, 8)) * 1e-20
    g2.quantities["density_2"] = np.random.random((4, 6, 8)) * 1e-20
    m = Model()
    m.set_grid(g)
    m.add_density_grid(g["density"], test_dust())
    m.add_density_grid(g["density_2"], test_dust())
    s = m.add_point_source_collection()
    s.luminosity, s.temperature = np.array([1., 2., 0.5]) * lsun, 7000.
    s.position = np.array([[0.1, 0.2, 0.3], [-0.5, 0.4, -0.2], [0.6, -0.6, 0.1]]) * pc
    s = m.add_external_box_source()
    s.luminosity, s.temperature, s.bounds = 3 * lsun, 3000., [[-pc, pc], [-pc, pc], [-pc, pc]]
    image(m, track_origin="basic", stokes=False)
    m.set_n_initial_iterations(4)
    m.set_convergence(True, percentile=99., absolute=3., relative=2.)
    m.set_n_photons(initial=4000, imaging=3000)
    m.conf.output.output_specific_energy = "all"
    m.conf.output.output_density = "last"
    m.set_seed(-102)
    keep(m, "native_amr.rtin", tmp)


def spherical(tmp):
    """spherical polar; luminosity-map source with the 'lte' spectrum is not writable by the front-end -> map + temperature;
    modified random walk; kill_on_absorb off; enforce_energy_range off"""
    np.random.seed(3)
    grid = SphericalPolarGrid(np.linspace(0., 3. * pc, 6), np.linspace(0., np.pi, 8), np.linspace(0., 2. * np.pi, 4))
    m = Model()
    m.set_grid(grid)
    m.add_density_grid(np.random.random(grid.shape) * 4e-20, test_dust())
    s = m.add_map_source()
    s.luminosity, s.temperature, s.map = 2 * lsun, 4000., np.random.random(grid.shape)
    s = m.add_point_source()
    s.luminosity, s.temperature = lsun, 8000.
    image(m, views=((60., 10.),), n=5)
    m.set_n_initial_iterations(2)
    m.set_mrw(True, gamma=2., inter_max=100)
    m.set_enforce_energy_range(False)
    m.set_n_photons(initial=4000, imaging=3000)
    m.set_seed(-103)
    keep(m, "native_sph.rtin", tmp)


def cylindrical(tmp):
    """cylindrical polar; external spherical source + plane-parallel beam; monochromatic final iteration + raytracing"""
    np.random.seed(4)
    grid = CylindricalPolarGrid(np.linspace(0., 2. * pc, 8), np.linspace(-pc, pc, 4), np.linspace(0., 2. * np.pi, 6))
    m = Model()
    m.set_grid(grid)
    m.add_density_grid(np.random.random(grid.shape) * 2e-20, test_dust())
    s = m.add_external_spherical_source()
    s.luminosity, s.temperature, s.radius, s.position = 2 * lsun, 5000., 0.9 * pc, (0., 0., 0.)
    s = m.add_point_source()
    s.luminosity, s.temperature, s.position = lsun, 7000., (0.2 * pc, 0.1 * pc, 0.)
    m.set_monochromatic(True, wavelengths=[1., 10., 100.])
    i = m.add_peeled_images(sed=True, image=True)
    i.set_viewing_angles([30., 80.], [20., 300.])
    i.set_image_size(5, 5)
    i.set_image_limits(-2 * pc, 2 * pc, -2 * pc, 2 * pc)
    i.set_aperture_range(2, 0.5 * pc, 3. * pc)
    i.set_track_origin("scatterings", n_scat=2)
    m.set_raytracing(True)
    m.set_n_initial_iterations(2)
    m.set_n_photons(initial=4000, imaging_sources=2000, imaging_dust=2000, raytracing_sources=2000, raytracing_dust=2000)
    m.set_seed(-104)
    keep(m, "native_cyl.rtin", tmp)


def voronoi(tmp):
    """Voronoi (voro++ through the front-end), 60 sites; point source; binned images (no forced first interaction)"""
    np.random.seed(5)
    n = 60
    x, y, z = (np.random.uniform(-pc, pc, n) for _ in range(3))
    grid = VoronoiGrid(x, y, z)
    m = Model()
    m.set_grid(grid)
    m.add_density_grid(np.random.random(n) * 3e-20, test_dust())
    s = m.add_point_source()
    s.luminosity, s.temperature, s.position = lsun, 6000., (0.05 * pc, 0.02 * pc, -0.03 * pc)
    b = m.add_binned_images(sed=True, image=True)
    b.set_viewing_bins(3, 4)
    b.set_image_size(4, 4)
    b.set_image_limits(-1.5 * pc, 1.5 * pc, -1.5 * pc, 1.5 * pc)
    b.set_aperture_range(1, 2 * pc, 2 * pc)
    b.set_wavelength_range(3, 0.1, 1000.)
    image(m, views=((90., 0.),), n=4)
    m.set_forced_first_interaction(False)
    m.set_n_initial_iterations(2)
    m.set_n_photons(initial=4000, imaging=4000)
    m.set_seed(-105)
    keep(m, "native_vor.rtin", tmp)


if __name__ == "__main__":
    which = sys.argv[1:] or ["octree", "amr", "spherical", "cylindrical", "voronoi"]
    with tempfile.TemporaryDirectory() as tmp:
        for w in which:
            globals()[w](tmp)
