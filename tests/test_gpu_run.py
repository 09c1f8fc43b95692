"""The run() counterpart end to end on the GPU: iteration sequencing, outputs,
failure convention, and the file-level drop-in (.rtin in, .rtout out)."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

import hyperion_amd
from cases import GOLDEN, golden_problem
from hyperion_amd.benchmark import PC, make_benchmark_problem
from hyperion_amd.run import run, run_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONDA = "/opt/conda/bin/python3.9"


def test_iteration_sequence_and_output_modes():
    prob, z = golden_problem("car_peeloff.False.npz")
    prob.config.n_initial_photons = 20000
    prob.config.n_last_photons = 20000
    prob.config.output_specific_energy = "all"
    r = run_problem(prob)
    assert [it.index for it in r.iterations] == [1, 2, 3, 4, 5] and r.n_iterations == 5 and not r.converged
    assert all(it.specific_energy is not None and it.killed_geo == 0 for it in r.iterations)
    assert r.peeled[1]["seds"].shape == z["golden/group2/seds"].shape
    assert r.peeled[2]["images"].shape == z["golden/group3/images"].shape
    # SED apertures are cumulative after normalisation
    s = r.peeled[0]["seds"][0]
    assert np.all(np.diff(s, axis=2) >= -1e-12 * s.max())
    # total absorbed luminosity in line with the golden (which used 1e3 packets: ~5 % noise)
    w = prob.density * prob.volumes
    assert (r.iterations[-1].specific_energy * w).sum() == pytest.approx((z["golden/specific_energy_last"] * w).sum(), rel=0.15)
    prob.config.output_specific_energy = "last"
    r2 = run_problem(prob)
    assert [it.specific_energy is not None for it in r2.iterations] == [False] * 4 + [True]
    np.testing.assert_allclose(r2.iterations[-1].specific_energy, r.iterations[-1].specific_energy, rtol=1e-10)


def test_raytracing_run_sequence():
    """main.f90:255-305 with raytracing on: Lucy iterations, final iteration peeling scattered packets
    only, raytracing iteration; the golden's SED of the raytraced run within its Monte Carlo noise."""
    prob, z = golden_problem("car_peeloff_ray.False.npz")
    prob.config.n_initial_photons = 20000
    prob.config.n_last_photons = 40000
    prob.config.n_ray_photons_sources = 20000
    prob.config.n_ray_photons_dust = 30000
    r = run_problem(prob)
    gold = z["golden/group2/seds"]
    got = r.peeled[1]["seds"]
    assert got.shape == gold.shape
    # direct source light (origin slot 0) is noise-free up to source sampling: within 5 % of the golden (2000 rays)
    assert got[0, 0, 0, -1, :].sum() == pytest.approx(gold[0, 0, 0, -1, :].sum(), rel=0.08)
    assert got[0, 1].max() > 0 and got[0, 2].max() > 0
    assert r.raytracing_stats["killed_geo"] == 0


def test_convergence_stops_the_iterations():
    p = make_benchmark_problem(8, n_photons=200000, n_iter=10)
    p.config.check_convergence = True
    p.config.convergence_absolute = 2.0
    p.config.convergence_relative = 2.0
    p.config.convergence_percentile = 90.0
    p.config.output_specific_energy = "last"
    r = run_problem(p)
    assert r.converged and r.n_iterations < 10
    assert r.iterations[-1].index == r.n_iterations and r.iterations[-1].specific_energy is not None


def test_npz_round_trip_and_failure_convention(tmp_path):
    p = make_benchmark_problem(8, n_photons=20000, n_iter=2)
    src = str(tmp_path / "in.npz")
    p.to_npz(src)
    out = run(src, str(tmp_path / "out.npz"), overwrite=True, logfile=str(tmp_path / "log.txt"))
    z = np.load(out)
    assert z["iteration_00002/specific_energy"].shape == (1, 8, 8, 8) and int(z["iterations"]) == 2
    assert "starting Lucy iteration 2" in open(tmp_path / "log.txt").read()
    with pytest.raises(SystemExit, match="already exists"):
        run(src, out)
    # a source outside the grid: message in the log, SystemExit with the reference's text
    p.sources[0].position = (5 * PC, 0, 0)
    p.to_npz(src)
    with pytest.raises(SystemExit, match="An error occurred, and the run did not complete"):
        run(src, out, overwrite=True, logfile=str(tmp_path / "log2.txt"))
    assert "photon was not emitted inside a cell" in open(tmp_path / "log2.txt").read()
    assert not os.path.exists(out)
    # a mode the input asks for and the engine cannot honour is refused, never ignored: PDA with a version-1 dust file
    # (setup_rt.f90:289-300)
    p, _ = golden_problem("car_specific_energy.False.False.npz")
    p.config.pda = True
    with pytest.raises(hyperion_amd.EngineError, match="version 1 dust files can no longer be used when PDA is computed"):
        run_problem(p)


@pytest.mark.skipif(not os.path.exists(CONDA), reason="no python with h5py in this image")
def test_file_level_drop_in_rtin_to_rtout(tmp_path):
    """`hyperion [-f] in.rtin out.rtout` counterpart on a .rtin written by the
    reference front-end; success == the output carries date_ended
    (scripts/hyperion:98-104)."""
    out = str(tmp_path / "model.rtout")
    env = dict(os.environ, PYTHONPATH=ROOT)
    rc = subprocess.call([CONDA, "-W", "ignore", "-m", "hyperion_amd", "-f", os.path.join(GOLDEN, "car_peeloff.False.rtin"), out],
                         env=env, cwd=ROOT)
    assert rc == 0
    code = ("import h5py, numpy as np\n"
            "f = h5py.File(%r, 'r')\n"
            "assert f.attrs['date_ended'] and f.attrs['iterations'] == 5\n"
            "assert f['iteration_00005/specific_energy'].shape == (1, 3, 5, 7)\n"
            "assert f['Peeled/group_00003/images'].shape == (4, 12, 1, 6, 6, 4)\n"
            "assert float(np.nansum(f['Peeled/group_00001/seds'][0])) > 0\n") % out
    subprocess.check_call([CONDA, "-W", "ignore", "-c", code])
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):
        shutil.copy(out, os.path.join(keep, "car_peeloff.False.gpu.rtout"))


@pytest.mark.parametrize("tau", ["0.1", "1"])
def test_monochromatic_run_matches_the_pascucci_golden(tau, tmp_path):
    """program main with `monochromatic` on (main.f90:271-272): Lucy iterations, do_final_mono,
    do_raytracing, exact-frequency normalisation nu F_nu; compared with the reference's
    test_pascucci.tau=*.rtout (one 1000-packet realisation) at 100 x the packets: every
    (view, wavelength) bin above 1 % of the peak within 25 % (the golden's own coherent noise is
    ~4 % from the raytraced stellar sphere plus the scattered / thermal shot noise), the
    wavelength-summed flux of each view within 8 %, and the .rtout carries the frequencies table."""
    from hyperion_amd.run import write_rtout
    prob, z = golden_problem("pascucci.tau=%s.npz" % tau)
    c = prob.config
    c.n_initial_photons = 100000
    c.n_last_photons_sources = c.n_last_photons_dust = 100000
    c.n_ray_photons_sources = c.n_ray_photons_dust = 100000
    r = run_problem(prob)
    seds = r.peeled[0]["seds"]
    gold = z["golden/seds"]
    assert seds.shape == gold.shape == (4, 1, 3, 1, 61)
    I, g = seds[0, 0, :, 0, :], gold[0, 0, :, 0, :]
    sel = I > 1e-2 * I.max()
    assert sel.sum() > 60
    dev = np.abs(g[sel] / I[sel] - 1.0)
    # (the optically thick discs, tau = 10 and 100, are pinned at equal packet numbers by the oracle's z-score test:
    # their 5 x 1000-packet temperature structure is too noisy for a comparison against a converged run)
    assert np.percentile(dev, 90) < 0.15 and dev.max() < 0.25
    for iv in range(3):
        assert g[iv].sum() == pytest.approx(I[iv].sum(), rel=0.08)
    try:
        import h5py          # the system python of the GPU image has none; /opt/conda's does
    except ImportError:
        return
    out = str(tmp_path / "pascucci.rtout")
    write_rtout(out, prob, r)
    with h5py.File(out, "r") as f:
        grp = f["Peeled/group_00001"]
        np.testing.assert_allclose(grp["frequencies"][...]["nu"], prob.config.frequencies, rtol=1e-15)
        assert "numin" not in grp["seds"].attrs and "apmin" in grp["seds"].attrs


def test_filters_run_reproduces_the_reference_known_answer():
    """hyperion/model/tests/test_filters.py through run_problem (n_initial_iter = 0: only the final iteration)."""
    from test_oracle_features import check_filter_known_answers
    prob, z = golden_problem("car_filters.npz")
    prob.config.n_last_photons = 200000
    r = run_problem(prob)
    assert r.n_iterations == 0 and r.peeled[0]["images"].shape == (1, 1, 3, 20, 10, 2) and r.peeled[0]["seds"].shape == (1, 1, 3, 1, 2)
    check_filter_known_answers(prob, z, r.peeled[0]["images"])
    # the SED aperture takes in the whole sky, the image only its frame: SED >= image summed over the pixels, and close to it
    sed, img = r.peeled[0]["seds"][0, 0, :, 0, :], r.peeled[0]["images"][0, 0].sum(axis=(1, 2))
    assert np.all(sed >= img * (1 - 1e-12)) and np.all(sed <= img * 1.02)


@pytest.mark.skipif(not os.path.exists(CONDA), reason="no python with h5py in this image")
def test_every_output_switch_of_the_boundary_is_honoured(tmp_path):
    """A reference-written .rtin (tests/golden/car_options.rtin, made by make_fixtures.py filters) that turns on what
    src/main/setup_rt.f90:247-283, src/grid/grid_generic.f90:29-130, src/images/image_type.f90:173-181,690-777 and
    src/main/main.f90:133-150 read: filters + 4-byte cubes in one image group, n_photons / density / density_diff /
    specific_energy_spectrum datasets, 4-byte grid datasets, copy_input."""
    out = str(tmp_path / "options.rtout")
    env = dict(os.environ, PYTHONPATH=ROOT)
    rc = subprocess.call([CONDA, "-W", "ignore", "-m", "hyperion_amd", "-f", os.path.join(GOLDEN, "car_options.rtin"), out], env=env, cwd=ROOT)
    assert rc == 0
    code = ("import h5py, numpy as np\n"
            "f = h5py.File(%r, 'r')\n"
            "assert f.attrs['date_ended'] and f.attrs['iterations'] == 2\n"
            "assert isinstance(f.get('Input', getlink=True), h5py.HardLink) and 'Grid' in f['Input'] and f['Input'].attrs['copy_input'] == b'yes'\n"
            "g1, g2 = f['iteration_00001'], f['iteration_00002']\n"
            "assert sorted(g1) == ['n_photons', 'specific_energy'], sorted(g1)\n"
            "assert sorted(g2) == ['density', 'density_diff', 'n_photons', 'specific_energy', 'specific_energy_spectrum', 'specific_energy_spectrum_bin_edges'], sorted(g2)\n"
            "assert g2['specific_energy'].dtype == np.float32 and g2['density'].dtype == np.float32 and g2['specific_energy'].shape == (1, 3, 5, 7)\n"
            "assert g2['n_photons'].shape == (3, 5, 7) and g2['n_photons'].dtype.kind == 'i' and g2['n_photons'][...].max() > 100\n"
            "assert not g2['density_diff'][...].any()\n"
            "sp = g2['specific_energy_spectrum'][...]; assert sp.shape == (6, 1, 3, 5, 7) and sp.dtype == np.float32\n"
            "np.testing.assert_allclose(g2['specific_energy_spectrum_bin_edges'][...], np.logspace(10., 16., 7), rtol=1e-12)\n"
            "se = g2['specific_energy'][...]; ok = sp.sum(axis=0) > 0\n"
            "assert ok.mean() > 0.9 and np.all(sp.sum(axis=0)[ok] <= se[ok] * 1.0001)\n"
            "p1, p2 = f['Peeled/group_00001'], f['Peeled/group_00002']\n"
            "assert p1.attrs['use_filters'] == b'yes' and p1.attrs['n_filt'] == 1 and p1['filt_nu0'].shape == (1,)\n"
            "assert p1['images'].dtype == np.float32 and p1['images'].shape == (1, 1, 2, 5, 4, 1) and 'numin' not in p1['images'].attrs\n"
            "assert p2['images'].dtype == np.float64 and 'numin' in p2['images'].attrs and p2['seds'].shape == (1, 1, 1, 2, 4)\n"
            "assert float(p1['images'][...].sum()) > 0 and float(p2['seds'][...].sum()) > 0\n") % out
    subprocess.check_call([CONDA, "-W", "ignore", "-c", code])
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):
        shutil.copy(out, os.path.join(keep, "car_options.gpu.rtout"))
