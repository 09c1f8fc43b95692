"""Multi-GPU sharding of one iteration: one process per GPU, packets split into
contiguous global-id ranges, ONE all-reduce of the accumulator block.

Replaces the reference's MPI master/worker chunk dispatcher and
``MPI_Reduce`` + ``MPI_Bcast`` (``src/mpi/mpi_routines.f90:62-323``): because the
per-packet Philox streams are keyed by the global packet id, the result does not
depend on the number of ranks (up to FP64 summation order), and because every
rank applies ``update_energy_abs`` to the same reduced block no broadcast is
needed.

Error agreement: a device error (frequency outside the dust table, packet emitted
outside the grid, negative t in find_wall) can hit one rank only, because it
depends on that rank's packet ids.  The reference stops every rank through
``error()``; here the flag travels IN the data collective: a spare slot of the
block's scalar tail (``TAIL_RANK_ERROR``) is 1 on a rank in error, the sum is
non-zero on every rank, and ``finish`` raises everywhere -- nobody waits in
``all_reduce`` for a rank that has raised, and there is no second collective.
"""
from __future__ import annotations


def shard_range(n_total, rank, world_size):
    """Contiguous id range [first, first + n_local) of `rank`."""
    first = (n_total * rank) // world_size
    last = (n_total * (rank + 1)) // world_size
    return first, last - first


def _try(fn, *args, **kw):
    """Run a launch; return the exception instead of raising so that the ranks can agree on it."""
    try:
        fn(*args, **kw)
    except Exception as e:
        return e
    return None


def _collective(engine, name, world_size, all_reduce, force, error=None, agree=None, timing=None):
    """Sum the accumulator block of iteration kind `name` ('lucy', 'final', 'raytracing', 'mono') over the ranks.
    `engine.<name>_accumulators_tensor()` returns the tensor alias of the device block (it waits for the kernels), or
    raises the engine's error; `error` is an exception the launch already raised on this rank.

    ONE collective: where the engine names a spare slot in the block's scalar tail
    (`engine.flag_index(name)`), a rank in error adds 1 there (to a zero block of the same length if it has none) and
    raises after the sum; the `finish` call of the other ranks finds the slot non-zero and raises too.  Adapters without
    that slot agree through `agree(flag) -> max over ranks` first.  `timing` (dict) receives seconds spent waiting for the
    kernels and in the collective."""
    import time
    if world_size == 1 and all_reduce is None and not force:
        if error is not None:
            raise error
        return False
    flag_index = engine.flag_index(name) if hasattr(engine, "flag_index") else None
    if all_reduce is not None and flag_index is None and agree is None and world_size > 1:
        # an adapter with its own all_reduce, no spare slot in the block and no `agree`: a rank in error could only raise
        # locally and leave its peers in all_reduce.  Refused on EVERY rank, before anybody enters the collective (the
        # configuration is the same everywhere, so all ranks raise this together)
        raise ValueError("sharded %s iteration over %d ranks: an engine adapter without flag_index() needs `agree` "
                         "(max of an int over the ranks) so that the ranks can agree on an error" % (name, world_size))
    t0 = time.perf_counter()
    err = error
    acc = None
    if err is None:
        try:
            acc = getattr(engine, name + "_accumulators_tensor")()
        except Exception as e:        # EngineError of this rank: tell the others before raising
            err = e
    t1 = time.perf_counter()
    if flag_index is not None:
        if acc is None:
            acc = engine.zero_block(name)
        if err is not None:
            acc[flag_index] += 1.0
    elif all_reduce is not None:
        if agree is not None:
            if agree(1 if err is not None else 0):
                raise err if err is not None else RuntimeError("another rank reported an engine error")
        elif err is not None:
            raise err       # world size 1 (refused above otherwise): nobody else is waiting
    else:
        import torch
        import torch.distributed as dist
        flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            raise err if err is not None else RuntimeError("another rank reported an engine error")
    if all_reduce is None:
        import torch
        import torch.distributed as dist
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        # the engine reads the block on its own HIP stream: wait for RCCL's stream
        if acc.is_cuda:
            torch.cuda.synchronize()
    else:
        all_reduce(acc)
    if timing is not None:
        timing["t_kernel_s"] = t1 - t0
        timing["t_collective_s"] = time.perf_counter() - t1
    if err is not None:
        raise err
    return True


def lucy_iteration_sharded(engine, n_total, iteration, rank=0, world_size=1, all_reduce=None, want_output=True,
                           force_collective=False, agree=None):
    """One Lucy iteration of `n_total` packets over `world_size` ranks.

    `engine` provides lucy_launch / lucy_accumulators_tensor / lucy_finish
    (hyperion_amd.Engine); `all_reduce(tensor)` sums in place across ranks
    (default ``torch.distributed.all_reduce``; backend "nccl" is RCCL over xGMI).
    The returned stats carry where this rank's time went: t_launch_s (host side of the launches), t_kernel_s (waiting
    for the propagation), t_collective_s, t_finish_s."""
    import time
    first, n_local = shard_range(n_total, rank, world_size)
    t0 = time.perf_counter()
    err = _try(engine.lucy_launch, first, n_local, iteration)
    t1 = time.perf_counter()
    timing = {}
    if not _collective(engine, "lucy", world_size, all_reduce, force_collective, error=err, agree=agree, timing=timing):
        engine.lucy_accumulators()           # no collective: no torch needed
        timing = {"t_kernel_s": time.perf_counter() - t1, "t_collective_s": 0.0}
    t2 = time.perf_counter()
    out, stats = engine.lucy_finish(want_output=want_output)
    stats["n_packets"] = n_total
    stats["t_launch_s"] = t1 - t0
    stats.update(timing)
    stats["t_finish_s"] = time.perf_counter() - t2
    return out, stats


def final_iteration_sharded(engine, n_total, rank=0, world_size=1, all_reduce=None, force_collective=False, agree=None):
    first, n_local = shard_range(n_total, rank, world_size)
    err = _try(engine.final_launch, first, n_local)
    if not _collective(engine, "final", world_size, all_reduce, force_collective, error=err, agree=agree):
        engine.final_accumulators()
    res, stats = engine.final_finish()
    stats["n_packets"] = n_total
    return res, stats


def raytracing_iteration_sharded(engine, n_sources, n_dust, rank=0, world_size=1, all_reduce=None, force_collective=False, agree=None):
    """do_raytracing (src/main/iter_raytracing.f90) sharded by packet id over the ranks.  Rank 0
    keeps the cubes of the final iteration, the others start from zero, so that ONE all-reduce of
    the image block yields final + raytraced flux on every rank."""
    err = None
    for which, n_total in ((0, n_sources), (1, n_dust)):
        first, n_local = shard_range(n_total, rank, world_size)
        err = err or _try(engine.raytracing_launch, which, first, n_local, n_total, zero_first=(which == 0 and rank > 0))
    _collective(engine, "raytracing", world_size, all_reduce, force_collective, error=err, agree=agree)
    res, stats = engine.raytracing_finish()
    stats["n_packets"] = n_sources + n_dust
    return res, stats


def mono_iteration_sharded(engine, n_sources, n_dust, n_frequencies, rank=0, world_size=1, all_reduce=None, force_collective=False,
                           agree=None):
    """do_final_mono (src/main/iter_final_mono.f90) sharded by packet id: every rank runs its id range of every
    (part, frequency) launch into cubes it zeroed itself, then ONE all-reduce of the image block (cubes and
    counters) as after the polychromatic final iteration."""
    first_launch = True
    err = None
    for which, n_total in ((0, n_sources), (1, n_dust)):
        first, n_local = shard_range(n_total, rank, world_size)
        for inu in range(n_frequencies):
            if err is None:
                err = _try(engine.mono_launch, which, inu, first, n_local, n_total, zero_first=first_launch)
            first_launch = False
    _collective(engine, "mono", world_size, all_reduce, force_collective, error=err, agree=agree)
    res, stats = engine.mono_finish()
    stats["n_packets"] = (n_sources + n_dust) * n_frequencies
    return res, stats
