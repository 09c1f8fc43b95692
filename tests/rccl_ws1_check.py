"""Run under ``python -m torch.distributed.run --nproc-per-node 1``: the sharded Lucy, final, raytracing and
monochromatic iterations with a REAL ``torch.distributed.all_reduce`` (backend "nccl" = RCCL) on the zero-copy alias
of the engine's device block, compared with the unsharded calls of a second engine.  Exercises the
``__cuda_array_interface__`` alias and the hand-over between RCCL's stream and the engine's own streams on hardware
(src/mpi/mpi_routines.f90:272-323 is what the collective replaces).  Prints RCCL_WS1_OK on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    import hyperion_amd
    from hyperion_amd.benchmark import make_benchmark_problem
    from hyperion_amd.distributed import (final_iteration_sharded, lucy_iteration_sharded, mono_iteration_sharded,
                                          raytracing_iteration_sharded)
    from cases import golden_problem

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
    # 1. Lucy iteration, both schedules (persistent kernel and brick-tiled generations on three streams)
    prob = make_benchmark_problem(32, n_photons=300000, n_iter=2)
    for mode in (0, 1):
        a, b = hyperion_amd.Engine(prob, device=0), hyperion_amd.Engine(prob, device=0)
        a.set_option("lucy_mode", mode); b.set_option("lucy_mode", mode)
        for it in (1, 2):
            sa, ta = lucy_iteration_sharded(a, 300000, it, rank, world, force_collective=True)
            sb, tb = b.lucy_iteration(300000, it)
            assert ta["crossings"] == tb["crossings"] and ta["interactions"] == tb["interactions"], (mode, ta, tb)
            np.testing.assert_allclose(sa, sb, rtol=1e-12, atol=0)
        a.close(); b.close()
    # 2. imaging: final + raytracing iterations on the reference's peel-off model
    prob, _ = golden_problem("car_peeloff_ray.False.npz")
    a, b = hyperion_amd.Engine(prob, device=0), hyperion_amd.Engine(prob, device=0)
    for e in (a, b):
        e.lucy_iteration(20000, 1)
    ra, ta = final_iteration_sharded(a, 20000, rank, world, force_collective=True)
    rb, tb = b.final_iteration(20000)
    for x, y in zip(ra, rb):
        for k in x:
            np.testing.assert_allclose(x[k], y[k], rtol=1e-12, atol=0)
    ra, _ = raytracing_iteration_sharded(a, 20000, 20000, rank, world, force_collective=True)
    rb, _ = b.raytracing_iteration(20000, 20000)
    for x, y in zip(ra, rb):
        for k in x:
            np.testing.assert_allclose(x[k], y[k], rtol=1e-12, atol=0)
    a.close(); b.close()
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_WS1_OK", flush=True)


if __name__ == "__main__":
    main()
