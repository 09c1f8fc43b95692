"""Multi-GPU sharding of one iteration: one process per GPU, packets split into
contiguous global-id ranges, ONE all-reduce of the accumulator block.

Replaces the reference's MPI master/worker chunk dispatcher and
``MPI_Reduce`` + ``MPI_Bcast`` (``src/mpi/mpi_routines.f90:62-323``): because the
per-packet Philox streams are keyed by the global packet id, the result does not
depend on the number of ranks (up to FP64 summation order), and because every
rank applies ``update_energy_abs`` to the same reduced block no broadcast is
needed.
"""
from __future__ import annotations


def shard_range(n_total, rank, world_size):
    """Contiguous id range [first, first + n_local) of `rank`."""
    first = (n_total * rank) // world_size
    last = (n_total * (rank + 1)) // world_size
    return first, last - first


def lucy_iteration_sharded(engine, n_total, iteration, rank=0, world_size=1, all_reduce=None, want_output=True,
                           force_collective=False):
    """One Lucy iteration of `n_total` packets over `world_size` ranks.

    `engine` provides lucy_launch / lucy_accumulators_tensor / lucy_finish
    (hyperion_amd.Engine); `all_reduce(tensor)` sums in place across ranks
    (``torch.distributed.all_reduce``; backend "nccl" is RCCL over xGMI)."""
    first, n_local = shard_range(n_total, rank, world_size)
    engine.lucy_launch(first, n_local, iteration)
    if world_size == 1 and all_reduce is None and not force_collective:
        engine.lucy_accumulators()           # no collective: no torch needed
        acc = None
    else:
        acc = engine.lucy_accumulators_tensor()
    if world_size > 1 or force_collective:
        if all_reduce is None:
            import torch
            import torch.distributed as dist
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
            # the engine reads the block on its own HIP stream: wait for RCCL's stream
            torch.cuda.synchronize()
        else:
            all_reduce(acc)
    out, stats = engine.lucy_finish(want_output=want_output)
    stats["n_packets"] = n_total
    return out, stats


def final_iteration_sharded(engine, n_total, rank=0, world_size=1, all_reduce=None):
    first, n_local = shard_range(n_total, rank, world_size)
    engine.final_launch(first, n_local)
    if world_size == 1 and all_reduce is None:
        engine.final_accumulators()
        acc = None
    else:
        acc = engine.final_accumulators_tensor()
    if world_size > 1:
        if all_reduce is None:
            import torch
            import torch.distributed as dist
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
        else:
            all_reduce(acc)
    res, stats = engine.final_finish()
    stats["n_packets"] = n_total
    return res, stats


def raytracing_iteration_sharded(engine, n_sources, n_dust, rank=0, world_size=1, all_reduce=None):
    """do_raytracing (src/main/iter_raytracing.f90) sharded by packet id over the ranks.  Rank 0
    keeps the cubes of the final iteration, the others start from zero, so that ONE all-reduce of
    the image block yields final + raytraced flux on every rank."""
    for which, n_total in ((0, n_sources), (1, n_dust)):
        first, n_local = shard_range(n_total, rank, world_size)
        engine.raytracing_launch(which, first, n_local, n_total, zero_first=(which == 0 and rank > 0))
    if world_size > 1:
        acc = engine.raytracing_accumulators_tensor()
        if all_reduce is None:
            import torch
            import torch.distributed as dist
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
        else:
            all_reduce(acc)
    res, stats = engine.raytracing_finish()
    stats["n_packets"] = n_sources + n_dust
    return res, stats


def mono_iteration_sharded(engine, n_sources, n_dust, n_frequencies, rank=0, world_size=1, all_reduce=None):
    """do_final_mono (src/main/iter_final_mono.f90) sharded by packet id: every rank runs its id range of every
    (part, frequency) launch into cubes it zeroed itself, then ONE all-reduce of the image block (cubes and
    counters) as after the polychromatic final iteration."""
    first_launch = True
    for which, n_total in ((0, n_sources), (1, n_dust)):
        first, n_local = shard_range(n_total, rank, world_size)
        for inu in range(n_frequencies):
            engine.mono_launch(which, inu, first, n_local, n_total, zero_first=first_launch)
            first_launch = False
    if world_size > 1:
        acc = engine.mono_accumulators_tensor()
        if all_reduce is None:
            import torch
            import torch.distributed as dist
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
        else:
            all_reduce(acc)
    res, stats = engine.mono_finish()
    stats["n_packets"] = (n_sources + n_dust) * n_frequencies
    return res, stats
