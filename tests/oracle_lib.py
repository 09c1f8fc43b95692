"""ctypes wrapper of the CPU oracle (oracle/hyp_oracle.c).  Test infrastructure:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it."""
import ctypes as C
import os
import subprocess

import numpy as np

from hyperion_amd._abi import IterStats, MarshalledProblem, ProblemDesc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "build", "libhyp_oracle.so")

_dp = C.POINTER(C.c_double)


def build_oracle(force=False):
    if os.environ.get("HYP_ORACLE_SO"):       # tools/mutation_check.py: a deliberately broken oracle, to show that a test has power
        return os.environ["HYP_ORACLE_SO"]
    src = os.path.join(ORACLE_DIR, "hyp_oracle.c")
    hdr = os.path.join(ORACLE_DIR, "hyp_oracle.h")
    stale = (not os.path.exists(ORACLE_SO)
             or (os.path.exists(src) and max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(ORACLE_SO)))
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"] + (["-B"] if force else []))
    return ORACLE_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build_oracle())
        L.orc_create.argtypes = [C.POINTER(ProblemDesc), C.POINTER(C.c_void_p)]
        L.orc_create.restype = C.c_int
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_destroy.restype = None
        L.orc_last_error.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_global_error.restype = C.c_char_p
        L.orc_lucy_iteration.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, _dp, C.POINTER(IterStats)]
        L.orc_lucy_accumulate.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(IterStats)]
        L.orc_lucy_finish.argtypes = [C.c_void_p, _dp, C.POINTER(IterStats)]
        L.orc_set_accumulators.argtypes = [C.c_void_p, _dp]
        for f in ("orc_specific_energy_sum", "orc_specific_energy", "orc_density"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = _dp
        L.orc_n_photons.argtypes = [C.c_void_p]
        L.orc_n_photons.restype = C.POINTER(C.c_int64)
        for f in ("orc_specific_energy_spectrum", "orc_specific_energy_sum_spectrum"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = _dp
        L.orc_pda_last_cells.argtypes = [C.c_void_p]
        L.orc_convergence_value.argtypes = [C.c_void_p, _dp, C.c_double, _dp]
        L.orc_final_iteration.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(IterStats)]
        L.orc_final_accumulate.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(IterStats)]
        L.orc_final_scale.argtypes = [C.c_void_p, C.c_double]
        L.orc_raytracing_iteration.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(IterStats)]
        L.orc_raytracing_accumulate.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(IterStats)]
        L.orc_mono_iteration.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(IterStats)]
        L.orc_mono_accumulate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(IterStats)]
        L.orc_peeled_sed_rw.argtypes = [C.c_void_p, C.c_int]
        L.orc_peeled_sed_rw.restype = _dp
        L.orc_peeled_img_rw.argtypes = [C.c_void_p, C.c_int]
        L.orc_peeled_img_rw.restype = _dp
        L.orc_peeled_n_orig.argtypes = [C.c_void_p, C.c_int]
        for f in ("orc_peeled_sed", "orc_peeled_img", "orc_peeled_sed2", "orc_peeled_img2"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
            getattr(L, f).restype = _dp
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_philox4x32_10.restype = None
        L.orc_walk_ray.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.orc_probe_uniform.argtypes = [C.c_int64, C.c_int, C.c_uint64, C.c_int]
        L.orc_probe_uniform.restype = C.c_double
        L.orc_probe_scatter.argtypes = [C.c_void_p, C.c_int, C.c_double, _dp, _dp, C.c_int64, C.c_uint64, _dp, _dp]
        L.orc_probe_scatter.restype = None
        L.orc_probe_sample_jnu.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_probe_sample_jnu.restype = C.c_double
        L.orc_probe_planck.argtypes = [C.c_double, C.c_int64, C.c_uint64]
        L.orc_probe_planck.restype = C.c_double
        L.orc_probe_rotate.argtypes = [_dp, _dp, _dp, _dp]
        L.orc_probe_rotate.restype = None
        L.orc_probe_optconsts.argtypes = [C.c_void_p, C.c_int, C.c_double, _dp]
        L.orc_probe_optconsts.restype = None
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


class Oracle:
    """CPU oracle for one Problem (same call sequence as hyperion_amd.Engine)."""

    def __init__(self, problem):
        self.m = MarshalledProblem(problem)
        self.problem = problem
        self.h = C.c_void_p()
        rc = lib().orc_create(C.byref(self.m.desc), C.byref(self.h))
        if rc != 0:
            raise OracleError(lib().orc_global_error().decode())
        self.shape = problem.density.shape

    def close(self):
        if self.h:
            lib().orc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self):
        return lib().orc_last_error(self.h).decode()

    def lucy_iteration(self, n_packets, iteration, n_threads=0):
        out = np.empty(self.shape, dtype=np.float64)
        st = IterStats()
        rc = lib().orc_lucy_iteration(self.h, n_packets, iteration, n_threads,
                                      out.ctypes.data_as(_dp), C.byref(st))
        if rc != 0:
            raise OracleError(self._err())
        return out, st.as_dict()

    def lucy_accumulate(self, first_id, n_local, iteration, n_threads=0):
        st = IterStats()
        rc = lib().orc_lucy_accumulate(self.h, first_id, n_local, iteration, n_threads, C.byref(st))
        if rc != 0:
            raise OracleError(self._err())
        n = int(np.prod(self.shape))
        s = np.ctypeslib.as_array(lib().orc_specific_energy_sum(self.h), shape=(n,)).reshape(self.shape).copy()
        return s, st.as_dict()

    def lucy_finish(self, block=None):
        """update_energy_abs on the pending (or the given, all-reduced) accumulator block."""
        if block is not None:
            block = np.ascontiguousarray(block, dtype=np.float64)
            lib().orc_set_accumulators(self.h, block.ctypes.data_as(_dp))
        out = np.empty(self.shape, dtype=np.float64)
        st = IterStats()
        rc = lib().orc_lucy_finish(self.h, out.ctypes.data_as(_dp), C.byref(st))
        if rc != 0:
            raise OracleError(self._err())
        return out, st.as_dict()

    def specific_energy(self):
        n = int(np.prod(self.shape))
        return np.ctypeslib.as_array(lib().orc_specific_energy(self.h), shape=(n,)).reshape(self.shape).copy()

    def density(self):
        n = int(np.prod(self.shape))
        return np.ctypeslib.as_array(lib().orc_density(self.h), shape=(n,)).reshape(self.shape).copy()

    def n_photons(self):
        """n_photons of the last Lucy iteration, shape of one species of the density."""
        p = lib().orc_n_photons(self.h)
        if not p:
            return None
        n = int(np.prod(self.shape[1:]))
        return np.ctypeslib.as_array(p, shape=(n,)).reshape(self.shape[1:]).copy()

    def specific_energy_spectrum(self, sums=False):
        """(n_bins, n_dust, cells...) frequency-resolved specific energy (or its raw sums)."""
        nb = int(self.m.desc.config.n_spectrum_bins)
        if nb == 0:
            return None
        fn = lib().orc_specific_energy_sum_spectrum if sums else lib().orc_specific_energy_spectrum
        n = int(np.prod(self.shape)) * nb
        return np.ctypeslib.as_array(fn(self.h), shape=(n,)).reshape((nb,) + tuple(self.shape)).copy()

    def pda_last_cells(self):
        return int(lib().orc_pda_last_cells(self.h))

    def convergence_value_against(self, prev, percentile):
        """(status, value) of specific_energy_converged's tested quantity against the given previous specific energy."""
        prev = np.ascontiguousarray(prev, dtype=np.float64)
        v = C.c_double()
        rc = lib().orc_convergence_value(self.h, prev.ctypes.data_as(_dp), float(percentile), C.byref(v))
        return int(rc), v.value

    def convergence_value(self, percentile):
        """Same call as Engine.convergence_value: against the specific energy at the previous call (status 3 the first time)."""
        cur = self.specific_energy()
        prev = getattr(self, "_prev_se", None)
        self._prev_se = cur
        if prev is None:
            return 3, 0.0
        return self.convergence_value_against(prev, percentile)

    def raytracing_iteration(self, n_sources, n_dust, n_threads=0):
        """do_raytracing: adds to the cubes of the last final_iteration; returns them."""
        st = IterStats()
        rc = lib().orc_raytracing_iteration(self.h, int(n_sources), int(n_dust), n_threads, C.byref(st))
        if rc != 0:
            raise OracleError(self._err())
        return self._peeled(), st.as_dict()

    def raytracing_accumulate(self, which, first_id, n_local, n_total, zero_first=False, n_threads=0):
        st = IterStats()
        rc = lib().orc_raytracing_accumulate(self.h, int(which), int(first_id), int(n_local), int(n_total), int(bool(zero_first)),
                                             n_threads, C.byref(st))
        if rc != 0:
            raise OracleError(self._err())
        return st.as_dict()

    def mono_iteration(self, n_sources, n_dust, n_threads=0):
        """do_final_mono: all frequencies, source then dust packets; returns the cubes."""
        st = IterStats()
        rc = lib().orc_mono_iteration(self.h, int(n_sources), int(n_dust), n_threads, C.byref(st))
        if rc != 0:
            raise OracleError(self._err())
        return self._peeled(), st.as_dict()

    def mono_accumulate(self, which, inu, first_id, n_local, n_total, zero_first=False, n_threads=0):
        st = IterStats()
        rc = lib().orc_mono_accumulate(self.h, int(which), int(inu), int(first_id), int(n_local), int(n_total), int(bool(zero_first)),
                                       n_threads, C.byref(st))
        if rc != 0:
            raise OracleError(self._err())
        return st.as_dict()

    def peeled_views(self):
        """Writable numpy views (sed, img) of every group's cubes."""
        out = []
        for g in range(len(self.problem.peeled) + (1 if self.problem.binned is not None else 0)):
            n_orig = lib().orc_peeled_n_orig(self.h, g)
            sed_shape, img_shape = self.m.peeled_shapes(g, n_orig)
            v = {}
            if sed_shape is not None:
                v["sed"] = np.ctypeslib.as_array(lib().orc_peeled_sed_rw(self.h, g), shape=(int(np.prod(sed_shape)),))
            if img_shape is not None:
                v["img"] = np.ctypeslib.as_array(lib().orc_peeled_img_rw(self.h, g), shape=(int(np.prod(img_shape)),))
            out.append(v)
        return out

    def final_iteration(self, n_packets, n_threads=0):
        st = IterStats()
        rc = lib().orc_final_iteration(self.h, n_packets, n_threads, C.byref(st))
        if rc != 0:
            raise OracleError(self._err())
        return self._peeled(), st.as_dict()

    def final_accumulate(self, first_id, n_local, n_threads=0):
        """Unscaled flux sums of the packet ids [first_id, first_id + n_local) into zeroed cubes (see peeled_views)."""
        st = IterStats()
        rc = lib().orc_final_accumulate(self.h, int(first_id), int(n_local), n_threads, C.byref(st))
        if rc != 0:
            raise OracleError(self._err())
        return st.as_dict()

    def final_scale(self, energy_current):
        lib().orc_final_scale(self.h, float(energy_current))

    def _peeled(self):
        out = []
        for g in range(len(self.problem.peeled) + (1 if self.problem.binned is not None else 0)):
            n_orig = lib().orc_peeled_n_orig(self.h, g)
            sed_shape, img_shape = self.m.peeled_shapes(g, n_orig)
            grp = {}
            for name, shape, fn, fn2 in (("sed", sed_shape, lib().orc_peeled_sed, lib().orc_peeled_sed2),
                                         ("img", img_shape, lib().orc_peeled_img, lib().orc_peeled_img2)):
                if shape is None:
                    continue
                n = int(np.prod(shape))
                grp[name] = np.ctypeslib.as_array(fn(self.h, g), shape=(n,)).reshape(shape).copy()
                grp[name + "2"] = np.ctypeslib.as_array(fn2(self.h, g), shape=(n,)).reshape(shape).copy()
            out.append(grp)
        return out

    def walk_ray(self, r0, v):
        r0 = np.ascontiguousarray(r0, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        path = C.c_double()
        n = lib().orc_walk_ray(self.h, r0.ctypes.data_as(_dp), v.ctypes.data_as(_dp), C.byref(path))
        return n, path.value


class SerialOracle(Oracle):
    """One OpenMP thread per call: for ensembles of small runs spread over a thread pool (ctypes drops the GIL)."""

    def lucy_iteration(self, n_packets, iteration, n_threads=1):
        return Oracle.lucy_iteration(self, n_packets, iteration, n_threads=1)

    def final_iteration(self, n_packets, n_threads=1):
        return Oracle.final_iteration(self, n_packets, n_threads=1)

    def raytracing_iteration(self, n_sources, n_dust, n_threads=1):
        return Oracle.raytracing_iteration(self, n_sources, n_dust, n_threads=1)

    def mono_iteration(self, n_sources, n_dust, n_threads=1):
        return Oracle.mono_iteration(self, n_sources, n_dust, n_threads=1)


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)
