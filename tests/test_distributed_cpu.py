"""N>1 path on CPU: two gloo ranks shard one Lucy iteration by packet-id range,
all-reduce the accumulator block once, and finish redundantly -- the result must
equal the single-process run (FP64 summation order only).  The per-rank work is
done by the CPU oracle behind the same three-call interface the HIP engine
exposes (launch / accumulators / finish), so this exercises exactly
hyperion_amd.distributed."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cases import golden_problem, ragged_grid_problem
from hyperion_amd.distributed import lucy_iteration_sharded
from oracle_lib import Oracle


class OracleAsEngine:
    """Adapter: the oracle behind Engine's sharded-iteration interface."""

    def __init__(self, prob):
        self.o = Oracle(prob)
        self.n = int(np.prod(prob.density.shape))

    def lucy_launch(self, first, n_local, iteration):
        self.s, self.st = self.o.lucy_accumulate(first, n_local, iteration, n_threads=2)

    def lucy_accumulators_tensor(self):
        blk = np.concatenate([self.s.ravel(), [self.st["energy_current"], self.st["killed_geo"], self.st["killed_int"],
                                               self.st["crossings"], self.st["interactions"], 0, 0, 0]])
        self.t = torch.from_numpy(blk)
        return self.t

    def lucy_finish(self, want_output=True):
        return self.o.lucy_finish(self.t.numpy())

    # raytracing iteration: the image block is the concatenation of all cubes
    def raytracing_launch(self, which, first, n_local, n_total, zero_first=False):
        self.o.raytracing_accumulate(which, first, n_local, n_total, zero_first, n_threads=2)

    def raytracing_accumulators_tensor(self):
        self.views = [v[k] for v in self.o.peeled_views() for k in ("sed", "img") if k in v]
        self.blk = torch.from_numpy(np.concatenate(self.views))
        return self.blk

    def raytracing_finish(self):
        off = 0
        for v in self.views:
            v[:] = self.blk.numpy()[off:off + v.size]
            off += v.size
        return self.o._peeled(), {"killed_geo": 0, "killed_int": 0}

    # polychromatic final iteration (mp_collect_images, mpi_routines.f90:381-459): unscaled cubes + sums of squares + the tail
    def final_launch(self, first, n_local):
        self.fst = self.o.final_accumulate(first, n_local, n_threads=2)

    def final_accumulators_tensor(self):
        from oracle_lib import lib
        import ctypes as C
        self.views = []
        for g, v in enumerate(self.o.peeled_views()):
            for k, fn2 in (("sed", lib().orc_peeled_sed2), ("img", lib().orc_peeled_img2)):
                if k in v:
                    self.views += [v[k], np.ctypeslib.as_array(fn2(self.o.h, g), shape=(v[k].size,))]
        tail = np.array([self.fst[k] for k in ("energy_current", "killed_geo", "killed_int", "crossings", "interactions")], dtype=np.float64)
        self.blk = torch.from_numpy(np.concatenate(self.views + [tail, np.zeros(3)]))
        return self.blk

    def final_finish(self):
        blk = self.blk.numpy()
        off = 0
        for v in self.views:
            v[:] = blk[off:off + v.size]
            off += v.size
        tail = blk[off:off + 8]
        if tail[5] != 0:
            raise RuntimeError("another rank reported an engine error")
        self.o.final_scale(tail[0])
        return self.o._peeled(), {"energy_current": float(tail[0]), "killed_geo": int(tail[1]), "killed_int": int(tail[2]),
                                  "crossings": int(tail[3]), "interactions": int(tail[4])}

    # monochromatic final iteration: same image block
    def mono_launch(self, which, inu, first, n_local, n_total, zero_first=False):
        self.o.mono_accumulate(which, inu, first, n_local, n_total, zero_first, n_threads=2)

    mono_accumulators_tensor = raytracing_accumulators_tensor
    mono_finish = raytracing_finish


def _agree(flag):
    """max of an int over the ranks: how an adapter without a spare slot in its block lets the ranks agree on an error"""
    t = torch.tensor([int(flag)], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, which, n_total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = golden_problem("car_specific_energy.False.True.npz")[0] if which == "kmh3" else ragged_grid_problem()
    eng = OracleAsEngine(prob)
    res = []
    for it in (1, 2):
        se, st = lucy_iteration_sharded(eng, n_total, it, rank, world, all_reduce=lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM), agree=_agree)
        res.append(se)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), se=np.array(res), crossings=st["crossings"], energy=st["energy_current"],
             n=st["n_packets"])
    dist.destroy_process_group()


@pytest.mark.parametrize("which", ["kmh3", "ragged"])
def test_two_rank_sharded_iteration_equals_single_process(tmp_path, which):
    n_total = 20001          # odd: uneven shards
    mp.spawn(_worker, args=(2, _free_port(), which, n_total, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["se"], r1["se"])          # every rank ends with the same state
    assert int(r0["n"]) == n_total
    prob = golden_problem("car_specific_energy.False.True.npz")[0] if which == "kmh3" else ragged_grid_problem()
    o = Oracle(prob)
    for k, it in enumerate((1, 2)):
        se, st = o.lucy_iteration(n_total, it, n_threads=2)
        np.testing.assert_allclose(r0["se"][k], se, rtol=1e-12)
    assert int(r0["crossings"]) == st["crossings"]
    assert float(r0["energy"]) == pytest.approx(st["energy_current"], rel=1e-14)


def _ray_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hyperion_amd.distributed import raytracing_iteration_sharded
    prob = golden_problem("car_peeloff_ray.False.npz")[0]
    eng = OracleAsEngine(prob)
    eng.o.lucy_iteration(3000, 1, n_threads=2)
    eng.o.final_iteration(2000, n_threads=2)          # every rank holds the same final cubes; rank 0's are kept
    res, st = raytracing_iteration_sharded(eng, 1501, 2001, rank, world, all_reduce=lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM), agree=_agree)
    np.savez(os.path.join(out_dir, "ray%d.npz" % rank), sed=res[1]["sed"], img=res[0]["img"])
    dist.destroy_process_group()


def test_two_rank_sharded_raytracing_equals_single_process(tmp_path):
    mp.spawn(_ray_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "ray0.npz"), np.load(tmp_path / "ray1.npz")
    np.testing.assert_array_equal(r0["sed"], r1["sed"])
    prob = golden_problem("car_peeloff_ray.False.npz")[0]
    o = Oracle(prob)
    o.lucy_iteration(3000, 1, n_threads=2)
    o.final_iteration(2000, n_threads=2)
    res, _ = o.raytracing_iteration(1501, 2001, n_threads=2)
    np.testing.assert_allclose(r0["sed"], res[1]["sed"], rtol=1e-12, atol=1e-14 * res[1]["sed"].max())
    np.testing.assert_allclose(r0["img"], res[0]["img"], rtol=1e-12, atol=1e-14 * res[0]["img"].max())


def _mono_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hyperion_amd.distributed import mono_iteration_sharded
    prob = golden_problem("pascucci.tau=1.npz")[0]
    eng = OracleAsEngine(prob)
    eng.o.lucy_iteration(2000, 1, n_threads=2)
    nf = prob.config.frequencies.size
    res, st = mono_iteration_sharded(eng, 301, 201, nf, rank, world, all_reduce=lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM), agree=_agree)
    np.savez(os.path.join(out_dir, "mono%d.npz" % rank), sed=res[0]["sed"])
    dist.destroy_process_group()


def test_two_rank_sharded_monochromatic_iteration_equals_single_process(tmp_path):
    """do_final_mono sharded by packet id over two gloo ranks: every (part, frequency) launch is split, one
    all-reduce of the image block at the end (hyperion_amd.distributed.mono_iteration_sharded)."""
    mp.spawn(_mono_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "mono0.npz"), np.load(tmp_path / "mono1.npz")
    np.testing.assert_array_equal(r0["sed"], r1["sed"])
    prob = golden_problem("pascucci.tau=1.npz")[0]
    o = Oracle(prob)
    o.lucy_iteration(2000, 1, n_threads=2)
    res, _ = o.mono_iteration(301, 201, n_threads=2)
    o.close()
    assert res[0]["sed"].max() > 0
    np.testing.assert_allclose(r0["sed"], res[0]["sed"], rtol=1e-12, atol=1e-14 * res[0]["sed"].max())


# --- error agreement inside the ONE collective: the flag rides in the tail of the block ------------------------------------

class FlaggedEngine(OracleAsEngine):
    """OracleAsEngine + the engine's spare tail slot for a rank's error (Engine.flag_index / zero_block) and, like
    hyp_lucy_finish, a finish that refuses when the summed slot is non-zero.  `fail` makes this rank's launch raise."""

    def __init__(self, prob, fail):
        super().__init__(prob)
        self.fail = fail
        self.n_collectives = 0

    def lucy_launch(self, first, n_local, iteration):
        if self.fail:
            raise RuntimeError("photon frequency is outside the range defined for the dust optical properties")
        super().lucy_launch(first, n_local, iteration)

    def flag_index(self, name):
        return self.n + 5            # TAIL_RANK_ERROR

    def zero_block(self, name):
        self.t = torch.zeros(self.n + 8, dtype=torch.float64)
        return self.t

    def lucy_finish(self, want_output=True):
        if self.t[self.n + 5] != 0:
            raise RuntimeError("another rank reported an engine error")
        return super().lucy_finish(want_output)


def _flag_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = FlaggedEngine(ragged_grid_problem(), fail=(rank == 1))

    def all_reduce(t):
        eng.n_collectives += 1
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    msg = "no error"
    try:
        lucy_iteration_sharded(eng, 4001, 1, rank, world, all_reduce=all_reduce)
    except RuntimeError as e:
        msg = str(e)
    with open(os.path.join(out_dir, "flag%d.txt" % rank), "w") as f:
        f.write("%d|%s" % (eng.n_collectives, msg))
    dist.destroy_process_group()


def test_error_on_one_rank_reaches_every_rank_through_the_one_collective(tmp_path):
    """Rank 1 fails in its launch: it still takes part in the all-reduce (a zero block with the flag slot set), raises its own
    error afterwards, and rank 0 -- whose launch was fine -- is told by the summed flag in finish.  One collective each."""
    mp.spawn(_flag_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = (tmp_path / "flag0.txt").read_text().split("|")
    r1 = (tmp_path / "flag1.txt").read_text().split("|")
    assert r0 == ["1", "another rank reported an engine error"]
    assert r1[0] == "1" and "outside the range defined" in r1[1]


# --- the polychromatic imaging iteration over two ranks ----------------------------------------------------------------------

class FlaggedImagingEngine(OracleAsEngine):
    """the image block with the engine's spare tail slot; `fail` makes this rank's launch raise"""

    def __init__(self, prob, fail):
        super().__init__(prob)
        self.fail = fail

    def final_launch(self, first, n_local):
        if self.fail:
            raise RuntimeError("photon was not emitted inside a cell")
        super().final_launch(first, n_local)

    def flag_index(self, name):
        return self._n_block() - 3

    def zero_block(self, name):
        self.views = []
        self.blk = torch.zeros(self._n_block(), dtype=torch.float64)
        return self.blk

    def _n_block(self):
        return 2 * sum(v[k].size for v in self.o.peeled_views() for k in ("sed", "img") if k in v) + 8


def _final_worker(rank, world, port, out_dir, fail_rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hyperion_amd.distributed import final_iteration_sharded
    prob = golden_problem("car_peeloff.False.npz")[0]
    eng = FlaggedImagingEngine(prob, fail=(rank == fail_rank))
    eng.o.lucy_iteration(3000, 1, n_threads=2)
    msg = "no error"
    try:
        res, st = final_iteration_sharded(eng, 4001, rank, world, all_reduce=lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM))
        np.savez(os.path.join(out_dir, "final%d.npz" % rank), sed=res[1]["sed"], sed2=res[1]["sed2"], img=res[0]["img"], crossings=st["crossings"],
                 energy=st["energy_current"], n=st["n_packets"])
    except RuntimeError as e:
        msg = str(e)
    with open(os.path.join(out_dir, "final%d.txt" % rank), "w") as f:
        f.write(msg)
    dist.destroy_process_group()


def test_two_rank_sharded_imaging_iteration_equals_single_process(tmp_path):
    """do_final over two gloo ranks (hyperion_amd.distributed.final_iteration_sharded): odd packet count, ONE all-reduce of
    [cubes | sums of squares | tail], scaling by the summed emitted energy on every rank."""
    mp.spawn(_final_worker, args=(2, _free_port(), str(tmp_path), -1), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "final0.npz"), np.load(tmp_path / "final1.npz")
    for k in ("sed", "sed2", "img"):
        np.testing.assert_array_equal(r0[k], r1[k])
    assert int(r0["n"]) == 4001
    prob = golden_problem("car_peeloff.False.npz")[0]
    o = Oracle(prob)
    o.lucy_iteration(3000, 1, n_threads=2)
    res, st = o.final_iteration(4001, n_threads=2)
    o.close()
    assert res[1]["sed"].max() > 0 and res[0]["img"].max() > 0
    np.testing.assert_allclose(r0["sed"], res[1]["sed"], rtol=1e-12, atol=1e-14 * res[1]["sed"].max())
    np.testing.assert_allclose(r0["sed2"], res[1]["sed2"], rtol=1e-12, atol=1e-14 * res[1]["sed2"].max())
    np.testing.assert_allclose(r0["img"], res[0]["img"], rtol=1e-12, atol=1e-14 * res[0]["img"].max())
    assert int(r0["crossings"]) == st["crossings"]
    assert float(r0["energy"]) == pytest.approx(st["energy_current"], rel=1e-13)


def test_error_in_the_imaging_iteration_of_one_rank_reaches_every_rank(tmp_path):
    mp.spawn(_final_worker, args=(2, _free_port(), str(tmp_path), 0), nprocs=2, join=True)
    assert "not emitted inside a cell" in (tmp_path / "final0.txt").read_text()
    assert (tmp_path / "final1.txt").read_text() == "another rank reported an engine error"


def test_an_adapter_that_cannot_report_errors_is_refused_at_world_size_two():
    """all_reduce given, no flag_index, no agree: every rank raises before anybody enters the collective"""
    from hyperion_amd.distributed import lucy_iteration_sharded as lis

    class Bare:
        def lucy_launch(self, *a):
            pass

    with pytest.raises(ValueError, match="agree"):
        lis(Bare(), 100, 1, rank=0, world_size=2, all_reduce=lambda t: None)
