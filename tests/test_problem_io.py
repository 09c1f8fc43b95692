"""Host logic: Problem validation, npz round trip, marshalling."""
import numpy as np
import pytest

from cases import golden_problem, ragged_grid_problem
from hyperion_amd._abi import MarshalledProblem
from hyperion_amd.benchmark import make_benchmark_problem
from hyperion_amd.distributed import shard_range
from hyperion_amd.problem import PeeledImages, Problem


def test_npz_round_trip(tmp_path):
    p, _ = golden_problem("car_peeloff.False.npz")
    path = str(tmp_path / "p.npz")
    p.to_npz(path)
    q = Problem.from_npz(path)
    assert q.shape == p.shape and q.n_dust == p.n_dust and len(q.sources) == 5 and len(q.peeled) == 3
    np.testing.assert_array_equal(q.density, p.density)
    np.testing.assert_array_equal(q.dust[0].P2, p.dust[0].P2)
    assert q.config == p.config
    assert q.peeled[1].track_origin == "basic" and q.peeled[2].track_origin == "detailed"
    assert q.peeled[0].n_view == 2 and q.peeled[0].n_wav == 5
    assert q.peeled[0].d_min == -np.inf and q.peeled[0].d_max == np.inf
    assert q.geometry_id == "a0af8431df2b52a1211bbd1b4ae67339"


def test_golden_fixture_contents():
    p, z = golden_problem("car_specific_energy.False.True.npz")
    assert p.shape == (7, 5, 3) and p.n_dust == 3
    assert z["golden/specific_energy"].shape == (5, 3, 3, 5, 7)
    d = p.dust[0]
    assert d.version == 1 and d.nu.size == 97 and d.mu.size == 100 and np.any(d.P2 != 0)
    assert d.emiss_jnu.shape == (817, 100)
    assert p.config.n_initial_photons == 10000 and p.config.n_initial_iter == 5


def test_validation_errors():
    p = make_benchmark_problem(4)
    with pytest.raises(ValueError, match="density array has wrong shape"):
        Problem(walls=p.walls, density=np.ones((1, 4, 4, 5)), dust=p.dust, sources=p.sources)
    with pytest.raises(ValueError, match="wrong number of dust types"):
        Problem(walls=p.walls, density=p.density, dust=p.dust, sources=p.sources, specific_energy=np.ones((2, 4, 4, 4)))
    p.config.forced_first_interaction_algorithm = "nope"
    with pytest.raises(ValueError, match="Unknown forced first interaction algorithm"):
        MarshalledProblem(p)
    p = make_benchmark_problem(4)
    p.sources[0].type = "banana"
    with pytest.raises(ValueError, match="unknown type in source list"):
        MarshalledProblem(p)


def test_marshalling_keeps_layout():
    p = ragged_grid_problem()
    m = MarshalledProblem(p)
    d = m.desc
    assert (d.grid.n1, d.grid.n2, d.grid.n3) == p.shape
    assert d.n_dust == 1 and d.n_sources == 2
    n = p.n_cells
    dens = np.ctypeslib.as_array(d.density, shape=(n,))
    np.testing.assert_array_equal(dens, p.density.ravel())
    assert d.sources[1].temperature == 5000.0
    assert d.dust[0].n_enu == p.dust[0].emiss_nu.size
    pl = PeeledImages(theta=[45.0], phi=[30.0], wav_min=1.0, wav_max=100.0)
    assert pl.nu_min == pytest.approx(2.99792458e12) and pl.nu_max == pytest.approx(2.99792458e14)


def test_shard_ranges_cover_everything_once():
    for n in (0, 1, 7, 1000, 10**9 + 7):
        for w in (1, 2, 3, 8):
            got = [shard_range(n, r, w) for r in range(w)]
            assert got[0][0] == 0
            for (f0, c0), (f1, _) in zip(got, got[1:]):
                assert f0 + c0 == f1
            assert got[-1][0] + got[-1][1] == n
            assert max(c for _, c in got) - min(c for _, c in got) <= 1
