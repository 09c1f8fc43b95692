"""The native driver end to end on the GPU: `hyperion_<grid> -f in.rtin out.rtout` (hyp_run.cpp over the C ABI, libhdf5, no
Python) against the Python adapter (`python -m hyperion_amd`) on the same reference-written inputs -- the same engine with the
same seeds underneath, so every dataset of the two .rtout files must agree to summation order and every attribute exactly --
one input per grid geometry, together covering the source types, filters, 4-byte outputs, copy_input, convergence exit, the
modified random walk, the monochromatic + raytracing sequence and binned images."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "hyperion_amd", "bin")
CONDA = "/opt/conda/bin/python3.9"
CASES = [("car", "car_peeloff.False.rtin"), ("car", "car_options.rtin"), ("oct", "native_oct.rtin"), ("amr", "native_amr.rtin"),
         ("sph", "native_sph.rtin"), ("cyl", "native_cyl.rtin"), ("vor", "native_vor.rtin")]

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.exists(os.path.join(BIN, "hyperion_amd_run")), reason="native driver not built (no C libhdf5)"),
              pytest.mark.skipif(not os.path.exists(CONDA), reason="no python with h5py to read the outputs back")]

COMPARE = r"""
import sys, h5py, numpy as np
a, b = h5py.File(sys.argv[1], "r"), h5py.File(sys.argv[2], "r")
skip_attrs = {"date_started", "date_ended", "cpu_time"}
def attrs(o):
    return {k: (v.decode() if isinstance(v, bytes) else v) for k, v in o.attrs.items() if k not in skip_attrs}
def names(f):
    out = []
    def visit(n, o):
        if not n.startswith("Input"):
            out.append(n)
    for k in f:
        if k == "Input":
            continue
        out.append(k)
        if isinstance(f[k], h5py.Group):
            f[k].visititems(lambda n, o, k=k: out.append(k + "/" + n))
    return sorted(out)
na, nb = names(a), names(b)
assert na == nb, (sorted(set(na) ^ set(nb)))
ra, rb = attrs(a), attrs(b)
assert set(ra) == set(rb), set(ra) ^ set(rb)
for k in ra:
    assert np.all(ra[k] == rb[k]), ("root attr", k, ra[k], rb[k])
assert "date_ended" in a.attrs and "date_ended" in b.attrs
n_data = 0
for n in na:
    x, y = a[n], b[n]
    xa, ya = attrs(x), attrs(y)
    assert set(xa) == set(ya), (n, set(xa) ^ set(ya))
    for k in xa:
        assert np.all(xa[k] == ya[k]), (n, k, xa[k], ya[k])
    if isinstance(x, h5py.Dataset):
        assert x.shape == y.shape and x.dtype == y.dtype, (n, x.shape, y.shape, x.dtype, y.dtype)
        u, v = x[...], y[...]
        if u.dtype.names:
            for fld in u.dtype.names:
                np.testing.assert_allclose(u[fld], v[fld], rtol=1e-14, err_msg=n)
        elif u.dtype.kind in "iu":
            assert np.array_equal(u, v), n
        else:
            tol = 2e-6 if u.dtype == np.float32 else 1e-9
            scale = float(np.nanmax(np.abs(v))) if v.size else 0.0
            np.testing.assert_allclose(u, v, rtol=tol, atol=1e-12 * scale, err_msg=n)
        n_data += 1
# the link to / copy of the input
la, lb = a.get("Input", getlink=True), b.get("Input", getlink=True)
assert type(la) == type(lb), (la, lb)
if isinstance(la, h5py.ExternalLink):
    assert la.filename == lb.filename and la.path == lb.path
else:
    assert sorted(a["Input"]) == sorted(b["Input"]) and set(a["Input"].attrs) == set(b["Input"].attrs)
print("compared", n_data, "datasets,", len(na), "objects")
"""


@pytest.mark.parametrize("suffix,name", CASES)
def test_native_driver_writes_what_the_python_adapter_writes(suffix, name, tmp_path):
    src = os.path.join(GOLDEN, name)
    native, ref = str(tmp_path / "native.rtout"), str(tmp_path / "python.rtout")
    r = subprocess.run([os.path.join(BIN, "hyperion_" + suffix), "-f", src, native], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    assert " [main] exiting final iteration" in r.stdout
    env = dict(os.environ, PYTHONPATH=ROOT)
    rc = subprocess.call([CONDA, "-W", "ignore", "-m", "hyperion_amd", "-f", src, ref], env=env, cwd=ROOT, timeout=600)
    assert rc == 0
    script = str(tmp_path / "compare.py")
    open(script, "w").write(COMPARE)
    out = subprocess.run([CONDA, "-W", "ignore", script, native, ref], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "compared" in out.stdout
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep) and name == "native_oct.rtin":
        import shutil
        shutil.copy(native, os.path.join(keep, "native_oct.native.rtout"))


def test_native_rtout_holds_physical_content_read_with_h5py(tmp_path):
    """The physical content of the native driver's output, read with plain h5py (not with the reference's ModelOutput: the reference
    does not travel to the GPU box; tools/validate_rtout_with_reference.py does that round trip in the build container): flux arrives
    in the SED, the specific energy is positive where there is dust, the apertures accumulate outwards."""
    native = str(tmp_path / "native.rtout")
    r = subprocess.run([os.path.join(BIN, "hyperion_oct"), "-f", os.path.join(GOLDEN, "native_oct.rtin"), native], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    code = ("import h5py, numpy as np\n"
            "f = h5py.File(%r, 'r')\n"
            "assert f.attrs['iterations'] == 2 and f.attrs['converged'] == b'no' and f.attrs['fortran_version']\n"
            "se = f['iteration_00002/specific_energy'][...]; assert se.shape == (1, 25) and (se > 0).sum() >= 20\n"
            "g = f['Peeled/group_00001']; s = g['seds'][...]; assert s.shape == (4, 6, 2, 2, 4) and s[0].sum() > 0\n"
            "assert g['seds_unc'].shape == s.shape and g['images'].shape == (4, 6, 2, 6, 6, 4) and g['images'].attrs['track_origin'] == b'detailed'\n"
            "assert np.all(np.diff(s[0].sum(axis=(0, 1, 3))) >= 0)\n") % native
    subprocess.check_call([CONDA, "-W", "ignore", "-c", code])


def test_mpi_name_runs_as_a_rank_and_equals_the_single_process_run(tmp_path):
    """`hyperion_car_mpi` as rank 0 of 1 (what a 1-GPU box can execute of `mpirun -n N hyperion_car_mpi`): the communicator is
    created from the launcher's environment, every iteration goes launch -> ncclAllReduce of the accumulator block -> finish,
    and the .rtout equals the one of the plain executable (a sum over one rank changes nothing).  Lucy + imaging + raytracing
    on the Cartesian peel-off model, monochromatic + raytracing on the spherical one."""
    for suffix, name in (("car", "car_peeloff.False.rtin"), ("sph", "native_sph.rtin")):
        src = os.path.join(GOLDEN, name)
        plain, ranked = str(tmp_path / (suffix + ".plain.rtout")), str(tmp_path / (suffix + ".rank.rtout"))
        r = subprocess.run([os.path.join(BIN, "hyperion_" + suffix), "-f", src, plain], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr + r.stdout
        env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        r = subprocess.run([os.path.join(BIN, "hyperion_" + suffix + "_mpi"), "-f", src, ranked], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr + r.stdout
        assert "[mpi] rank 0 of 1" in r.stdout and "RCCL all-reduce" in r.stdout
        script = str(tmp_path / "compare.py")
        open(script, "w").write(COMPARE)
        out = subprocess.run([CONDA, "-W", "ignore", script, ranked, plain], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-3000:]
    # the launcher form: --ranks 1 forks nothing and sets the same environment
    r = subprocess.run([os.path.join(BIN, "hyperion_car_mpi"), "-f", "--ranks", "1", os.path.join(GOLDEN, "car_peeloff.False.rtin"), str(tmp_path / "l.rtout")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "[mpi] rank 0 of 1" in r.stdout, r.stderr + r.stdout


def test_mpi_rank_refusing_to_start_tells_the_launcher(tmp_path):
    """An existing output without -f: rank 0 refuses before the rendezvous, with the reference's failure convention."""
    out = str(tmp_path / "exists.rtout")
    open(out, "w").write("x")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1")
    r = subprocess.run([os.path.join(BIN, "hyperion_car_mpi"), os.path.join(GOLDEN, "car_peeloff.False.rtin"), out], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "already exists" in r.stderr and "did not complete" in r.stderr


def test_a_rank_that_cannot_start_stops_its_peers_instead_of_leaving_them_in_the_rendezvous(tmp_path):
    """--ranks 2 on a box with ONE GPU: rank 1's hipSetDevice(1) fails before any communicator exists while rank 0 already
    waits in ncclCommInitRank for it.  Rank 1 leaves the launch's abort file, rank 0's watcher finds it and leaves with the
    reference's failure convention: status 1 within seconds instead of a hang, and no id / abort file stays behind."""
    import time
    code = "import torch, sys; sys.exit(0 if torch.cuda.device_count() == 1 else 3)"
    if subprocess.run(["python", "-c", code]).returncode != 0:
        pytest.skip("needs exactly one visible GPU")
    out = str(tmp_path / "two.rtout")
    t0 = time.time()
    r = subprocess.run([os.path.join(BIN, "hyperion_car_mpi"), "-f", "--ranks", "2", os.path.join(GOLDEN, "car_peeloff.False.rtin"), out],
                       capture_output=True, text=True, timeout=280, cwd=str(tmp_path))
    assert r.returncode == 1, r.stdout + r.stderr
    assert "hipSetDevice(1) failed" in r.stderr and "did not complete" in r.stderr
    assert time.time() - t0 < 120
    time.sleep(2.0)
    assert [f for f in os.listdir(tmp_path) if "ncclid" in f or "abort" in f or f.endswith(".tmp")] == []


def test_batch_system_variables_do_not_make_a_plain_executable_a_rank(tmp_path):
    """`hyperion_oct` inside an sbatch allocation with --ntasks=4 is one plain process: it runs to the end alone."""
    env = dict(os.environ, SLURM_PROCID="0", SLURM_NTASKS="4", SLURM_LOCALID="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    out = str(tmp_path / "slurm.rtout")
    r = subprocess.run([os.path.join(BIN, "hyperion_oct"), "-f", os.path.join(GOLDEN, "native_oct.rtin"), out], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0 and "[mpi]" not in r.stdout, r.stdout + r.stderr
