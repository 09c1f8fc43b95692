"""The native .rtin -> .rtout driver (hyperion_amd/csrc/hyp_run.cpp, built as hyperion_amd/bin/hyperion_<grid>): what can be
checked without a GPU -- it parses every reference-written input the Python reader parses, to the same numbers, and it fails
the way the reference's binary does (message on stderr, non-zero status, no `date_ended` in the output)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "hyperion_amd", "bin")
DRIVER = os.path.join(BIN, "hyperion_amd_run")
CONDA = "/opt/conda/bin/python3.9"
FIXTURES = ["car_peeloff.False.rtin", "car_options.rtin", "native_oct.rtin", "native_amr.rtin", "native_sph.rtin", "native_cyl.rtin", "native_vor.rtin"]

pytestmark = pytest.mark.skipif(not os.path.exists(DRIVER), reason="native driver not built (no C libhdf5 in this image)")


def _gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def check_input(name):
    out = subprocess.check_output([DRIVER, "--check-input", os.path.join(GOLDEN, name)], text=True)
    first = out.splitlines()[0].split()
    second = out.splitlines()[1].split()
    d = dict(zip(first[0::2], first[1::2]))
    d.update(zip(second[0::2], second[1::2]))
    d["groups"] = [dict(zip(l.split()[2::2], l.split()[3::2])) for l in out.splitlines() if l.startswith("group ")]
    d["dust"] = [dict(zip(l.split()[2::2], l.split()[3::2])) for l in out.splitlines() if l.startswith("dust ")]
    return d


def test_the_launcher_names_of_the_reference_exist():
    """scripts/hyperion:44-92 runs `hyperion_<suffix> [-f] input output`, or `mpirun -n N hyperion_<suffix>_mpi ...` with -m N"""
    for suffix in ("car", "sph", "cyl", "oct", "amr", "vor"):
        for name in ("hyperion_" + suffix, "hyperion_" + suffix + "_mpi"):
            p = os.path.join(BIN, name)
            assert os.path.exists(p) and os.access(p, os.X_OK), p
    r = subprocess.run([DRIVER], capture_output=True, text=True)
    assert r.returncode == 2 and "Usage:" in r.stderr and "[-f] [--ranks N] input_file output_file" in r.stderr


def test_the_driver_links_rccl_for_its_one_collective():
    """The N > 1 path of the native driver is rccl.h directly (ncclAllReduce over xGMI), not torch: the executable depends on
    librccl and the HIP runtime, and on nothing of the oracle."""
    out = subprocess.run(["ldd", DRIVER], capture_output=True, text=True).stdout
    assert "librccl" in out and "libamdhip64" in out and "libhyperion_amd" in out and "oracle" not in out


@pytest.mark.skipif(not os.path.exists(CONDA), reason="no python with h5py in this image")
@pytest.mark.parametrize("name", FIXTURES)
def test_native_reader_agrees_with_the_python_reader(name):
    d = check_input(name)
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from hyperion_amd.rtin import read_rtin\n"
            "p = read_rtin(%r); c = p.config\n"
            "print(p.grid_type, p.n_cells, p.n_dust, len(p.sources), len(p.peeled), int(p.binned is not None), c.n_initial_iter, c.n_initial_photons,"
            " c.n_last_photons, repr(float(np.asarray(p.density, dtype=float).sum())), c.seed, int(c.pda), int(c.mrw), int(c.raytracing), int(c.monochromatic),"
            " c.output_specific_energy, c.physics_io_bytes)\n"
            "for g in list(p.peeled) + ([p.binned] if p.binned is not None else []):\n"
            "    print('G', g.n_view, g.n_wav, g.n_x, g.n_y, g.n_ap, g.track_origin, int(g.uncertainties), int(g.compute_stokes), int(bool(g.filters)), g.io_bytes)\n"
            "for du in p.dust:\n"
            "    print('D', du.nu.size, du.mu.size, du.emiss_var.size, du.emiss_nu.size, du.mo_specific_energy.size, repr(float(du.chi[0])))\n") % (ROOT, os.path.join(GOLDEN, name))
    out = subprocess.check_output([CONDA, "-W", "ignore", "-c", code], text=True).splitlines()
    t = out[0].split()
    keys = ["grid_type", "n_cells", "n_dust", "n_sources", "n_peeled", "binned", "n_initial_iter", "n_initial_photons", "n_last_photons"]
    for k, v in zip(keys, t[:9]):
        assert d[k] == v, (k, d[k], v)
    assert float(d["density_sum"]) == pytest.approx(float(t[9]), rel=1e-14)
    for k, v in zip(["seed", "pda", "mrw", "raytracing", "monochromatic", "output_specific_energy", "physics_io_bytes"], t[10:17]):
        assert d[k] == v, (k, d[k], v)
    groups = [l.split()[1:] for l in out if l.startswith("G ")]
    assert len(groups) == len(d["groups"])
    origin = {"no": "0", "basic": "1", "detailed": "2", "scatterings": "3"}
    for g, h in zip(groups, d["groups"]):
        is_binned = h is d["groups"][-1] and d["binned"] == "1"
        if not is_binned:
            assert h["n_view"] == g[0]
        assert [h["n_nu"], h["n_x"], h["n_y"], h["n_ap"]] == g[1:5] and h["track_origin"] == origin[g[5]]
        assert [h["uncertainties"], h["stokes"], h["filters"], h["io_bytes"]] == g[6:10]
    dust = [l.split()[1:] for l in out if l.startswith("D ")]
    for a, b in zip(dust, d["dust"]):
        assert [b["n_nu"], b["n_mu"], b["n_jnu"], b["n_enu"], b["n_e"]] == a[:5] and float(b["chi0"]) == float(a[5])


def test_failures_look_like_the_reference_binarys(tmp_path):
    """no input -> message + status 1; an existing output is not overwritten without -f; without a GPU the run stops at
    hyp_create with the engine's message and the output it had started carries no date_ended"""
    out = str(tmp_path / "x.rtout")
    r = subprocess.run([DRIVER, str(tmp_path / "missing.rtin"), out], capture_output=True, text=True)
    assert r.returncode == 1 and "File does not exist" in r.stderr and "did not complete" in r.stderr and not os.path.exists(out)
    open(out, "w").write("precious")
    r = subprocess.run([DRIVER, os.path.join(GOLDEN, "native_oct.rtin"), out], capture_output=True, text=True)
    assert r.returncode == 1 and "already exists" in r.stderr and open(out).read() == "precious"
    bad = str(tmp_path / "bad.rtin")
    open(bad, "w").write("not hdf5")
    r = subprocess.run([DRIVER, "-f", bad, out], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open input file" in r.stderr
    if _gpu():
        return
    r = subprocess.run([DRIVER, "-f", os.path.join(GOLDEN, "native_oct.rtin"), out], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device available" in r.stderr and "An error occurred, and the run did not complete" in r.stderr
    if os.path.exists(CONDA):
        code = "import h5py; f = h5py.File(%r, 'r'); assert 'date_started' in f.attrs and 'date_ended' not in f.attrs" % out
        subprocess.check_call([CONDA, "-W", "ignore", "-c", code])


@pytest.mark.skipif(not os.path.exists(CONDA), reason="no python with h5py in this image")
@pytest.mark.parametrize("name", FIXTURES)
def test_two_independent_marshallers_hand_over_the_same_problem(name):
    """hyp_problem_digest (a digest, in a canonical order, of every scalar and array a hyp_problem points to) of the SAME
    reference-written .rtin marshalled twice, by two pieces of code that share nothing: hyp_run.cpp's HDF5 reader in C++, and
    hyperion_amd.rtin.read_rtin + hyperion_amd._abi.MarshalledProblem in Python -- the marshalling that feeds both the engine
    and, in tests/oracle_lib.py, the oracle.  A field one side mis-marshals (order, dtype, units, layout) changes its word:
    [0] grid + density, [1] dust tables, [2] sources, [3] run configuration + image groups."""
    out = subprocess.check_output([DRIVER, "--check-input", os.path.join(GOLDEN, name)], text=True)
    native = [l.split()[1:] for l in out.splitlines() if l.startswith("digest ")][0]
    code = ("import sys, ctypes as C; sys.path.insert(0, %r)\n"
            "from hyperion_amd.rtin import read_rtin\n"
            "from hyperion_amd._abi import MarshalledProblem\n"
            "from hyperion_amd.engine import load_library\n"
            "m = MarshalledProblem(read_rtin(%r))\n"
            "out = (C.c_uint64 * 4)()\n"
            "assert load_library().hyp_problem_digest(C.byref(m.desc), C.byref(out)) == 0\n"
            "print(' '.join('%%016x' %% v for v in out))\n") % (ROOT, os.path.join(GOLDEN, name))
    py = subprocess.check_output([CONDA, "-W", "ignore", "-c", code], text=True).split()
    assert py == native, dict(zip(("grid", "dust", "sources", "config+images"), zip(py, native)))


def test_a_rank_that_fails_early_takes_the_launch_down_and_leaves_no_files(tmp_path):
    """error() / mp_stop of the reference: with --ranks 2 rank 0 refuses an existing output before any communicator exists.
    Its peer must not be left waiting for the unique-id file (it is told through the launch's abort file, or fails on its own
    without a GPU), the launcher's status is 1, and neither the abort file, the id file nor a stray '.tmp' survives."""
    import time
    out = str(tmp_path / "x.rtout")
    open(out, "w").write("precious")
    t0 = time.time()
    r = subprocess.run([DRIVER, "--ranks", "2", os.path.join(GOLDEN, "native_oct.rtin"), out], capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
    assert r.returncode == 1 and "already exists" in r.stderr
    assert time.time() - t0 < 60
    assert open(out).read() == "precious"
    time.sleep(2.0)          # the rank that wrote the abort file removes it after its grace period
    left = sorted(os.listdir(tmp_path))
    assert left == ["x.rtout"], left


def test_batch_system_variables_do_not_make_a_plain_executable_a_rank(tmp_path):
    """`hyperion_oct` inside an sbatch allocation (SLURM_PROCID / SLURM_NTASKS set for every process of the job) is ONE
    process: it must not wait for peers.  Without a GPU it stops at hyp_create like any single run, at once."""
    if _gpu():
        pytest.skip("the GPU suite runs the same check to completion (tests/test_gpu_native_driver.py)")
    env = dict(os.environ, SLURM_PROCID="0", SLURM_NTASKS="4", SLURM_LOCALID="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    out = str(tmp_path / "y.rtout")
    r = subprocess.run([os.path.join(BIN, "hyperion_oct"), os.path.join(GOLDEN, "native_oct.rtin"), out], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 1 and "no HIP device available" in r.stderr and "[mpi]" not in r.stdout
