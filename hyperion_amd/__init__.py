"""MI355X-native photon-packet engine for Hyperion-style dust radiative transfer.

Only the Monte Carlo hot path of the reference (Lucy iteration + peel-off
imaging, ``src/main/iter_lucy.f90`` / ``iter_final.f90`` and callees) is
implemented, as hand-written HIP behind the C-ABI of ``include/hyperion_amd.h``.
"""
from .problem import Dust, PeeledImages, Problem, RunConfig, Source, Spot  # noqa: F401
from .engine import Engine, EngineError, load_library  # noqa: F401
from .run import run, run_problem  # noqa: F401

__all__ = ["Dust", "PeeledImages", "Problem", "RunConfig", "Source", "Engine", "EngineError", "load_library", "run", "run_problem"]
