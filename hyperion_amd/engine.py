"""ctypes binding of the C-ABI in ``include/hyperion_amd.h`` (the HIP engine).

There is no CPU fallback: if the shared library is missing or no GPU is
present, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from ._abi import ABI_VERSION, IterStats, MarshalledProblem, ProblemDesc
from .build import LIB

_dp = C.POINTER(C.c_double)

EXPORTS = [
    "hyp_abi_version", "hyp_problem_digest", "hyp_create", "hyp_destroy", "hyp_last_error",
    "hyp_lucy_iteration", "hyp_lucy_launch", "hyp_lucy_accumulators", "hyp_lucy_finish",
    "hyp_final_iteration", "hyp_final_launch", "hyp_final_accumulators", "hyp_final_finish",
    "hyp_peeled_get", "hyp_peeled_n_orig",
    "hyp_get_specific_energy", "hyp_get_density", "hyp_set_specific_energy",
    "hyp_last_kernel_ms", "hyp_set_option", "hyp_get_option",
    "hyp_raytracing_iteration", "hyp_raytracing_launch", "hyp_raytracing_accumulators", "hyp_raytracing_finish",
    "hyp_mono_iteration", "hyp_mono_launch", "hyp_mono_accumulators", "hyp_mono_finish",
    "hyp_get_n_photons", "hyp_get_specific_energy_spectrum", "hyp_convergence_value",
]


class EngineError(RuntimeError):
    """Raised when the engine reports a non-zero status; carries the message
    the reference would have printed before stopping."""


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm ships its own libamdhip64.so.  Two HIP runtimes in one process do not
    share the GPU: whichever initialises second reports "No HIP GPUs are available".  If torch
    is installed but not imported yet, map ITS runtime first (without importing torch) so that
    this library binds to the same copy torch will use later; if torch is already imported its
    runtime is the one in the process and nothing needs doing."""
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library(path=None):
    """Load libhyperion_amd.so and declare the prototypes.  Raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("HYPERION_AMD_LIB") or LIB      # (the variable: tuning and diagnostic builds of tools/variants.py)
    if not os.path.exists(path):
        raise EngineError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % path)
    _share_hip_runtime_with_torch()
    L = C.CDLL(path)
    H = C.c_void_p
    L.hyp_abi_version.restype = C.c_int
    L.hyp_problem_digest.argtypes = [C.POINTER(ProblemDesc), C.POINTER(C.c_uint64 * 4)]
    L.hyp_create.argtypes = [C.POINTER(ProblemDesc), C.c_int, C.POINTER(H)]
    L.hyp_destroy.argtypes = [H]
    L.hyp_destroy.restype = None
    L.hyp_last_error.argtypes = [H]
    L.hyp_last_error.restype = C.c_char_p
    L.hyp_lucy_iteration.argtypes = [H, C.c_uint64, C.c_int, _dp, C.POINTER(IterStats)]
    L.hyp_lucy_launch.argtypes = [H, C.c_uint64, C.c_uint64, C.c_int]
    L.hyp_lucy_accumulators.argtypes = [H, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.hyp_lucy_finish.argtypes = [H, _dp, C.POINTER(IterStats)]
    L.hyp_final_iteration.argtypes = [H, C.c_uint64, C.POINTER(IterStats)]
    L.hyp_final_launch.argtypes = [H, C.c_uint64, C.c_uint64]
    L.hyp_final_accumulators.argtypes = [H, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.hyp_final_finish.argtypes = [H, C.POINTER(IterStats)]
    L.hyp_peeled_get.argtypes = [H, C.c_int, C.c_int, _dp, C.POINTER(C.c_uint64)]
    L.hyp_peeled_n_orig.argtypes = [H, C.c_int]
    L.hyp_get_specific_energy.argtypes = [H, _dp]
    L.hyp_get_density.argtypes = [H, _dp]
    L.hyp_set_specific_energy.argtypes = [H, _dp]
    L.hyp_last_kernel_ms.argtypes = [H, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.hyp_set_option.argtypes = [H, C.c_char_p, C.c_int64]
    L.hyp_get_option.argtypes = [H, C.c_char_p, C.POINTER(C.c_int64)]
    L.hyp_raytracing_iteration.argtypes = [H, C.c_uint64, C.c_uint64, C.POINTER(IterStats)]
    L.hyp_raytracing_launch.argtypes = [H, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
    L.hyp_mono_iteration.argtypes = [H, C.c_uint64, C.c_uint64, C.POINTER(IterStats)]
    L.hyp_mono_launch.argtypes = [H, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
    L.hyp_mono_accumulators.argtypes = [H, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.hyp_mono_finish.argtypes = [H, C.POINTER(IterStats)]
    L.hyp_raytracing_accumulators.argtypes = [H, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.hyp_raytracing_finish.argtypes = [H, C.POINTER(IterStats)]
    L.hyp_get_n_photons.argtypes = [H, _dp]
    L.hyp_get_specific_energy_spectrum.argtypes = [H, _dp, _dp]
    L.hyp_convergence_value.argtypes = [H, C.c_double, _dp, C.POINTER(C.c_int)]
    if L.hyp_abi_version() != ABI_VERSION:      # the struct mirrors of _abi.py are for one version of include/hyperion_amd.h
        raise EngineError("%s has ABI version %d, this binding was written for %d: rebuild the extension"
                          % (path, L.hyp_abi_version(), ABI_VERSION))
    if path == (os.environ.get("HYPERION_AMD_LIB") or LIB):
        _lib = L
    return L


class _DeviceBlock:
    """Exposes a raw device pointer through ``__cuda_array_interface__`` so that
    torch can alias it (zero-copy) for the RCCL all-reduce."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {
            "shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 3, "strides": None,
        }


class Engine:
    """One problem resident on one GPU.  Mirrors the reference's run sequence
    (``src/main/main.f90:167-234``): create -> N x lucy_iteration -> final."""

    def __init__(self, problem, device=0):
        self._lib = load_library()
        self._m = MarshalledProblem(problem)
        self.problem = problem
        self.shape = problem.density.shape
        self._h = C.c_void_p()
        rc = self._lib.hyp_create(C.byref(self._m.desc), int(device), C.byref(self._h))
        if rc != 0:
            raise EngineError(self._lib.hyp_last_error(None).decode())
        self.device = int(device)

    # -- plumbing ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.hyp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise EngineError(self._lib.hyp_last_error(self._h).decode())

    def set_option(self, name, value):
        self._check(self._lib.hyp_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int64(0)
        self._check(self._lib.hyp_get_option(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    # -- Lucy iteration ---------------------------------------------------------
    def lucy_iteration(self, n_packets, iteration, want_output=True):
        out = np.empty(self.shape, dtype=np.float64) if want_output else None
        st = IterStats()
        self._check(self._lib.hyp_lucy_iteration(self._h, int(n_packets), int(iteration),
                                                 out.ctypes.data_as(_dp) if out is not None else None, C.byref(st)))
        return out, st.as_dict()

    def lucy_launch(self, first_id, n_local, iteration):
        self._check(self._lib.hyp_lucy_launch(self._h, int(first_id), int(n_local), int(iteration)))

    def lucy_accumulators(self):
        """(device pointer, n_doubles) of the accumulator block; waits for the kernel."""
        p = C.c_void_p()
        n = C.c_uint64()
        self._check(self._lib.hyp_lucy_accumulators(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def lucy_accumulators_tensor(self):
        import torch
        ptr, n = self.lucy_accumulators()
        return torch.as_tensor(_DeviceBlock(ptr, n), device="cuda:%d" % self.device)

    # -- sharded iterations: the error flag of a rank rides in the block's scalar tail (hyperion_amd.distributed) --------
    def flag_index(self, name):
        """Index (in doubles) of the TAIL_RANK_ERROR slot in the accumulator block of iteration kind `name`."""
        return self.get_option("lucy_flag_index" if name == "lucy" else "image_flag_index")

    def zero_block(self, name):
        """What a rank without results contributes to the collective: zeros of the block's length."""
        import torch
        n = self.get_option("lucy_block_doubles" if name == "lucy" else "image_block_doubles")
        return torch.zeros(n, dtype=torch.float64, device="cuda:%d" % self.device)

    def lucy_finish(self, want_output=True):
        out = np.empty(self.shape, dtype=np.float64) if want_output else None
        st = IterStats()
        self._check(self._lib.hyp_lucy_finish(self._h, out.ctypes.data_as(_dp) if out is not None else None, C.byref(st)))
        return out, st.as_dict()

    # -- final iteration -------------------------------------------------------
    def final_iteration(self, n_packets):
        st = IterStats()
        self._check(self._lib.hyp_final_iteration(self._h, int(n_packets), C.byref(st)))
        return self.peeled_results(), st.as_dict()

    def final_launch(self, first_id, n_local):
        self._check(self._lib.hyp_final_launch(self._h, int(first_id), int(n_local)))

    def final_accumulators(self):
        p = C.c_void_p()
        n = C.c_uint64()
        self._check(self._lib.hyp_final_accumulators(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def final_accumulators_tensor(self):
        import torch
        ptr, n = self.final_accumulators()
        return torch.as_tensor(_DeviceBlock(ptr, n), device="cuda:%d" % self.device)

    def final_finish(self):
        st = IterStats()
        self._check(self._lib.hyp_final_finish(self._h, C.byref(st)))
        return self.peeled_results(), st.as_dict()

    # -- raytracing iteration ----------------------------------------------------
    def raytracing_iteration(self, n_sources, n_dust):
        """do_raytracing: adds direct and thermal emission to the cubes of the last final iteration."""
        st = IterStats()
        self._check(self._lib.hyp_raytracing_iteration(self._h, int(n_sources), int(n_dust), C.byref(st)))
        return self.peeled_results(), st.as_dict()

    def raytracing_launch(self, which, first_id, n_local, n_total, zero_first=False):
        self._check(self._lib.hyp_raytracing_launch(self._h, int(which), int(first_id), int(n_local), int(n_total), int(bool(zero_first))))

    def raytracing_accumulators_tensor(self):
        import torch
        p = C.c_void_p()
        n = C.c_uint64()
        self._check(self._lib.hyp_raytracing_accumulators(self._h, C.byref(p), C.byref(n)))
        return torch.as_tensor(_DeviceBlock(p.value, n.value), device="cuda:%d" % self.device)

    def raytracing_finish(self):
        st = IterStats()
        self._check(self._lib.hyp_raytracing_finish(self._h, C.byref(st)))
        return self.peeled_results(), st.as_dict()

    # -- monochromatic final iteration (iter_final_mono.f90) ---------------------------
    def mono_iteration(self, n_sources, n_dust):
        """do_final_mono: every frequency, source packets then dust packets; zeroes the cubes first."""
        st = IterStats()
        self._check(self._lib.hyp_mono_iteration(self._h, int(n_sources), int(n_dust), C.byref(st)))
        return self.peeled_results(), st.as_dict()

    def mono_launch(self, which, inu, first_id, n_local, n_total, zero_first=False):
        self._check(self._lib.hyp_mono_launch(self._h, int(which), int(inu), int(first_id), int(n_local), int(n_total), int(bool(zero_first))))

    def mono_accumulators_tensor(self):
        import torch
        p = C.c_void_p()
        n = C.c_uint64()
        self._check(self._lib.hyp_mono_accumulators(self._h, C.byref(p), C.byref(n)))
        return torch.as_tensor(_DeviceBlock(p.value, n.value), device="cuda:%d" % self.device)

    def mono_finish(self):
        st = IterStats()
        self._check(self._lib.hyp_mono_finish(self._h, C.byref(st)))
        return self.peeled_results(), st.as_dict()

    # -- n_photons, frequency-resolved specific energy, convergence (grid_generic.f90:40-93, grid_physics_3d.f90:637-689) --
    def n_photons(self):
        """n_photons of the last Lucy iteration (whole job once the block has been all-reduced), shape of one species."""
        out = np.empty(self.shape[1:], dtype=np.float64)
        self._check(self._lib.hyp_get_n_photons(self._h, out.ctypes.data_as(_dp)))
        return np.rint(out).astype(np.int64)

    def specific_energy_spectrum(self):
        """(spectrum [n_bins, n_dust, cells...], bin_edges [n_bins + 1])."""
        nb = int(self._m.desc.config.n_spectrum_bins)
        out = np.empty((nb,) + tuple(self.shape), dtype=np.float64)
        edges = np.empty(nb + 1, dtype=np.float64)
        self._check(self._lib.hyp_get_specific_energy_spectrum(self._h, out.ctypes.data_as(_dp), edges.ctypes.data_as(_dp)))
        return out, edges

    def convergence_value(self, percentile):
        """(status, value): the quantity specific_energy_converged tests, against the specific energy at the previous call."""
        v, st = C.c_double(), C.c_int()
        self._check(self._lib.hyp_convergence_value(self._h, float(percentile), C.byref(v), C.byref(st)))
        return int(st.value), float(v.value)

    def peeled_results(self):
        out = []
        for g in range(len(self.problem.peeled) + (1 if self.problem.binned is not None else 0)):
            n_orig = self._lib.hyp_peeled_n_orig(self._h, g)
            sed_shape, img_shape = self._m.peeled_shapes(g, n_orig)
            grp = {}
            for which, name, shape in ((0, "sed", sed_shape), (1, "sed2", sed_shape),
                                       (2, "img", img_shape), (3, "img2", img_shape)):
                if shape is None:
                    continue
                a = np.empty(shape, dtype=np.float64)
                n = C.c_uint64(a.size)
                self._check(self._lib.hyp_peeled_get(self._h, g, which, a.ctypes.data_as(_dp), C.byref(n)))
                grp[name] = a
            out.append(grp)
        return out

    # -- state -------------------------------------------------------------------
    def specific_energy(self):
        out = np.empty(self.shape, dtype=np.float64)
        self._check(self._lib.hyp_get_specific_energy(self._h, out.ctypes.data_as(_dp)))
        return out

    def density(self):
        out = np.empty(self.shape, dtype=np.float64)
        self._check(self._lib.hyp_get_density(self._h, out.ctypes.data_as(_dp)))
        return out

    def set_specific_energy(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape != self.shape:
            raise ValueError("specific_energy array has wrong shape")
        self._check(self._lib.hyp_set_specific_energy(self._h, a.ctypes.data_as(_dp)))

    def last_kernel_ms(self):
        a, b = C.c_float(), C.c_float()
        self._check(self._lib.hyp_last_kernel_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value
