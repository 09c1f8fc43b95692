"""Host logic of the run() counterpart that needs no GPU: convergence test,
.rtin reader and .rtout writer (HDF5 side through /opt/conda's python, which
has h5py; the system python does not)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from cases import GOLDEN, golden_problem
from hyperion_amd.run import ConvergenceCheck, IterationRecord, RunResult

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONDA = "/opt/conda/bin/python3.9"
needs_h5py = pytest.mark.skipif(not os.path.exists(CONDA), reason="no python with h5py in this image")


class _ValueSource:
    """Stands in for Engine.convergence_value (hyp_convergence_value, computed on the device): same status codes, the
    quantile as restated in the oracle (element of rank nint(p / 100 (n - 1)) of the sorted ratios; tests/test_oracle_features.py
    checks the C implementation, tests/test_gpu_features.py the device one)."""

    def __init__(self):
        self.prev, self.cur = None, None

    def set(self, a):
        self.cur = np.asarray(a, dtype=float)

    def convergence_value(self, percentile):
        prev, cur = self.prev, self.cur
        self.prev = cur.copy()
        if prev is None:
            return 3, 0.0
        if np.all(prev == cur):
            return 1, 0.0
        if np.all((prev == cur) | (prev == 0) | (cur == 0)):
            return 2, 0.0
        m = (prev > 0) & (cur > 0) & (prev != cur)
        r = np.sort(np.maximum(prev[m] / cur[m], cur[m] / prev[m]))
        return 0, float(r[int(np.floor(percentile / 100.0 * (r.size - 1) + 0.5))])


def test_convergence_follows_reference_rules():
    """grid_physics_3d.f90:637-689"""
    src = _ValueSource()

    def step(c, a):
        src.set(a)
        return c(src)

    c = ConvergenceCheck(absolute=2.0, relative=1.5, percentile=99.0)
    a = np.ones((1, 2, 2, 2))
    assert step(c, a) is False                       # first call only stores the state
    assert step(c, a * 1.5) is False                 # no previous value yet
    assert step(c, a * 1.5 * 1.2) is True            # 1.2 < 2 and 1.5/1.2 < 1.5
    assert step(c, a * 1.5 * 1.2) is True            # unchanged -> exact convergence
    src = _ValueSource()
    c = ConvergenceCheck(absolute=1.1, relative=1.5, percentile=99.0)
    step(c, a); step(c, a * 1.5)
    assert step(c, a * 1.5 * 1.2) is False           # value 1.2 above the absolute threshold
    src = _ValueSource()
    c = ConvergenceCheck(absolute=2.0, relative=1.5, percentile=99.0)
    z = np.zeros_like(a)
    step(c, z)
    assert step(c, a) is False                       # only zero -> non-zero changes: cannot check


@needs_h5py
def test_rtin_reader_matches_fixture(tmp_path):
    out = tmp_path / "p.npz"
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from hyperion_amd.rtin import read_rtin\n"
            "read_rtin(%r).to_npz(%r)\n") % (ROOT, os.path.join(GOLDEN, "car_peeloff.False.rtin"), str(out))
    subprocess.check_call([CONDA, "-W", "ignore", "-c", code])
    from hyperion_amd.problem import Problem
    p = Problem.from_npz(str(out))
    q, _ = golden_problem("car_peeloff.False.npz")
    assert p.config == q.config and p.geometry_id == q.geometry_id
    np.testing.assert_array_equal(p.density, q.density)
    for k in ("nu", "chi", "albedo", "P1", "P2", "emiss_jnu", "emiss_var", "mo_specific_energy"):
        np.testing.assert_array_equal(getattr(p.dust[0], k), getattr(q.dust[0], k))
    assert [s.temperature for s in p.sources] == [s.temperature for s in q.sources]
    assert [(g.n_x, g.n_y, g.n_ap, g.n_wav, g.track_origin) for g in p.peeled] == \
        [(g.n_x, g.n_y, g.n_ap, g.n_wav, g.track_origin) for g in q.peeled]


@needs_h5py
def test_rtout_writer_reproduces_the_golden_layout(tmp_path):
    """Object names, shapes and attribute types of a written .rtout equal those
    of the reference's golden output for the same model."""
    out = tmp_path / "o.rtout"
    code = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np, h5py
from hyperion_amd.problem import Problem
from hyperion_amd.run import IterationRecord, RunResult, write_rtout
p = Problem.from_npz(%r)
its = [IterationRecord(i, 0, 0) for i in range(1, 6)]
its[-1].specific_energy = np.ones(p.density.shape)
peeled = []
for g, n_orig in zip(p.peeled, (1, 4, 12)):
    peeled.append({"seds": np.zeros((4, n_orig, g.n_view, g.n_ap, g.n_wav)),
                   "images": np.zeros((4, n_orig, g.n_view, g.n_y, g.n_x, g.n_wav))})
r = RunResult(its, False, 5, peeled, {"killed_geo": 0, "killed_int": 0}, 1.0, "now", "later")
write_rtout(%r, p, r, input_path=%r)
lay = {"root_attrs": {}, "items": {}}
with h5py.File(%r, "r") as f:
    lay["root_attrs"] = {k: type(v).__name__ for k, v in f.attrs.items()}
    def visit(n, o):
        e = {"attrs": {k: type(v).__name__ for k, v in o.attrs.items()}}
        if isinstance(o, h5py.Dataset):
            e["shape"] = list(o.shape); e["dtype"] = str(o.dtype)
        lay["items"][n] = e
    for k in f:
        if k != "Input":
            visit(k, f[k])
            if isinstance(f[k], h5py.Group):
                f[k].visititems(lambda n, o, k=k: visit(k + "/" + n, o))
    assert f.get("Input", getlink=True).__class__.__name__ == "ExternalLink"
print(json.dumps(lay))
''' % (ROOT, os.path.join(GOLDEN, "car_peeloff.False.npz"), str(out), os.path.join(GOLDEN, "car_peeloff.False.rtin"), str(out))
    got = json.loads(subprocess.check_output([CONDA, "-W", "ignore", "-c", code]).decode().strip().split("\n")[-1])
    want = json.load(open(os.path.join(GOLDEN, "rtout_layout.car_peeloff.json")))
    assert got["root_attrs"] == want["root_attrs"]
    assert got["items"] == want["items"]


@needs_h5py
def test_amr_rtin_reader_and_rtout_layout(tmp_path):
    """AMR file contract: Grid/Geometry/level_NNNNN/grid_NNNNN attrs + one density dataset per
    grid (hyperion/grid/amr_grid.py:336-420), and iteration outputs per level/grid
    (golden: test_specific_energy.grid_type=amr.*.rtout)."""
    q, _ = golden_problem("amr_specific_energy.False.True.npz")
    rtin, out, rtout = tmp_path / "amr.rtin", tmp_path / "p.npz", tmp_path / "amr.rtout"
    qn = tmp_path / "q.npz"
    q.to_npz(str(qn))
    code = ("import sys, h5py, numpy as np; sys.path.insert(0, %r)\n"
            "from hyperion_amd.problem import Problem\n"
            "from hyperion_amd.rtin import read_rtin\n"
            "from hyperion_amd.run import write_rtout, RunResult, IterationRecord\n"
            "q = Problem.from_npz(%r)\n"
            "with h5py.File(%r, 'r') as fi, h5py.File(%r, 'w') as fo:\n"
            "    for k, v in fi.attrs.items(): fo.attrs[k] = v\n"
            "    for k in fi:\n"
            "        if k != 'Grid': fi.copy(k, fo)\n"
            "    for i in (2, 3): fo.copy(fo['Dust/dust_001'], 'Dust/dust_%%03d' %% i)\n"
            "    g = fo.create_group('Grid/Geometry'); qq = fo.create_group('Grid/Quantities')\n"
            "    g.attrs['grid_type'] = np.bytes_('amr'); g.attrs['geometry'] = np.bytes_('abc'); g.attrs['nlevels'] = 2\n"
            "    start = 0\n"
            "    for k, (lev, n, b) in enumerate(zip(q.amr_level, q.amr_n, q.amr_bounds)):\n"
            "        gl = g.require_group('level_%%05d' %% lev); gl.attrs['ngrids'] = 1\n"
            "        gg = gl.create_group('grid_00001')\n"
            "        for name, v in zip(('xmin', 'xmax', 'ymin', 'ymax', 'zmin', 'zmax'), b): gg.attrs[name] = v\n"
            "        for name, v in zip(('n1', 'n2', 'n3'), n): gg.attrs[name] = int(v)\n"
            "        nc = int(np.prod(n))\n"
            "        qq.create_group('level_%%05d/grid_00001' %% lev).create_dataset('density', data=q.density[:, start:start + nc].reshape(3, n[2], n[1], n[0]))\n"
            "        start += nc\n"
            "p = read_rtin(%r)\n"
            "p.to_npz(%r)\n"
            "res = RunResult(iterations=[IterationRecord(index=1, killed_geo=0, killed_int=0, specific_energy=p.density * 2.0)], converged=False,\n"
            "                n_iterations=1, peeled=[], final_stats={}, cpu_time=0.0, date_started='x', date_ended='y')\n"
            "write_rtout(%r, p, res)\n"
            "with h5py.File(%r, 'r') as f:\n"
            "    a = f['iteration_00001/level_00002/grid_00001/specific_energy'][...]\n"
            "    assert a.shape == (3, 20, 6, 4), a.shape\n"
            "    assert np.array_equal(a.reshape(3, -1), 2.0 * p.density[:, 192:])\n"
            "    assert f['iteration_00001/level_00001/grid_00001/specific_energy'].shape == (3, 4, 6, 8)\n"
            ) % (ROOT, str(qn), os.path.join(GOLDEN, "car_peeloff.False.rtin"), str(rtin), str(rtin), str(out), str(rtout), str(rtout))
    subprocess.check_call([CONDA, "-W", "ignore", "-c", code])
    from hyperion_amd.problem import Problem
    p = Problem.from_npz(str(out))
    assert p.grid_type == "amr" and p.n_cells == 8 * 6 * 4 + 4 * 6 * 20
    np.testing.assert_array_equal(p.amr_n, q.amr_n)
    np.testing.assert_array_equal(p.amr_level, q.amr_level)
    np.testing.assert_array_equal(p.amr_bounds, q.amr_bounds)
    np.testing.assert_array_equal(p.density, q.density)
