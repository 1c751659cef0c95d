"""ctypes binding of libsgpr_hip.so (C ABI: include/sgpr_hip.h).

The HIP library is the product: there is NO CPU fallback.  Loading fails loudly if the shared
object is missing, and every call raises `SgprError` on a non-zero status (e.g. no gfx950
device).  Nothing here imports the test oracle.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGPR_HIP_LIB") or os.path.join(_HERE, "libsgpr_hip.so")  # override: kernel experiments

OK, E_INVALID, E_NODEVICE, E_NOMODEL, E_SPECIES, E_NOT_PD, E_UNSUPPORTED, E_OVERFLOW = 0, -1, -2, -3, -4, -5, -6, -7


class SgprError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsgpr_hip error {code}: {msg}")
        self.code = code


_lib = None
_vp, _i32, _i64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double

# name -> (restype, argtypes); must list every symbol include/sgpr_hip.h declares
SIGNATURES = {
    "sgpr_last_error": (C.c_char_p, []),
    "sgpr_version": (C.c_int, []),
    "sgpr_device_count": (C.c_int, []),
    "sgpr_create": (C.c_int, [C.c_int, C.c_int, _dbl, _dbl, C.c_int, _vp, _vp, C.c_int, C.POINTER(_vp)]),
    "sgpr_destroy": (None, [_vp]),
    "sgpr_set_inducing": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "sgpr_get_kmm": (C.c_int, [_vp, _vp]),
    "sgpr_get_inducing_descriptors": (C.c_int, [_vp, _vp]),
    "sgpr_set_weights": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "sgpr_set_mean": (C.c_int, [_vp, _vp, _vp]),
    "sgpr_get_choli": (C.c_int, [_vp, _vp]),
    "sgpr_restore_weights": (C.c_int, [_vp, _vp]),
    "sgpr_solve": (C.c_int, [_vp, C.c_int, _vp, _vp, _dbl, _vp, _vp, _vp, _vp]),
    "sgpr_resolve": (C.c_int, [_vp, _dbl, _vp, _vp, _vp, _vp]),
    "sgpr_jitcholesky": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "sgpr_resolve_batch": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "sgpr_make_vscale": (C.c_int, [_vp, _vp]),
    "sgpr_data_force_mae": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "sgpr_kernel_rows": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgpr_kernel_columns": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "sgpr_data_push": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int]),
    "sgpr_data_pop": (C.c_int, [_vp, C.c_int]),
    "sgpr_data_clear": (C.c_int, [_vp]),
    "sgpr_data_info": (C.c_int, [_vp, _vp, _vp]),
    "sgpr_data_matvec": (C.c_int, [_vp, _vp, _vp]),
    "sgpr_data_get": (C.c_int, [_vp, _vp]),
    "sgpr_data_fit_stats": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "sgpr_get_kmm_diag": (C.c_int, [_vp, _vp]),
    "sgpr_get_kmm_rowsum": (C.c_int, [_vp, _vp]),
    "sgpr_data_solve": (C.c_int, [_vp, _vp, C.c_int, _dbl, _vp, _vp, _vp, _vp]),
    "sgpr_data_factor": (C.c_int, [_vp, _vp, C.c_int]),
    "sgpr_add_inducing": (C.c_int, [_vp, _i32, C.c_int, _vp, _vp]),
    "sgpr_remove_inducing": (C.c_int, [_vp, C.c_int]),
    "sgpr_select_inducing": (C.c_int, [_vp, C.c_int, _vp]),
    "sgpr_kernel_local": (C.c_int, [_vp, _i32, C.c_int, _vp, _vp, _vp, _vp]),
    "sgpr_compute": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "sgpr_compute_view": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "sgpr_bind_system": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int]),
    "sgpr_packed_len": (_i64, [C.c_int]),
    "sgpr_step_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "sgpr_step_dev_next": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "sgpr_md_begin": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl]),
    "sgpr_md_run": (C.c_int, [_vp, C.c_int, _vp, _dbl, C.c_int, _vp, _vp, _vp]),
    "sgpr_md_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int]),
    "sgpr_md_velocities": (C.c_int, [_vp, _vp]),
    "sgpr_md_end": (C.c_int, [_vp]),
    "sgpr_md_seed": (C.c_int, [_vp, C.c_uint64]),
    "sgpr_md_thermostat": (C.c_int, [_vp, C.c_int, _dbl, _dbl]),
    "sgpr_md_deviates": (C.c_int, [_vp, _i64, C.c_int, _vp]),
    "sgpr_sync_check": (C.c_int, [_vp, _vp]),
    "sgpr_comm_unique_id": (C.c_int, [_vp]),
    "sgpr_comm_init": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "sgpr_comm_destroy": (C.c_int, [_vp]),
    "sgpr_comm_allreduce": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp]),
    "sgpr_peer_export": (C.c_int, [_vp, C.c_int, C.c_int, _i64, _vp]),
    "sgpr_peer_attach": (C.c_int, [_vp, _vp]),
    "sgpr_peer_destroy": (C.c_int, [_vp]),
    "sgpr_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "sgpr_get_list_rebuilds": (C.c_int, [_vp, _vp]),
    "sgpr_solve_info": (C.c_int, [_vp, _vp, C.c_int]),
    "sgpr_stress_from_virial": (C.c_int, [_vp, _vp, _vp]),
    "sgpr_get_descriptors": (C.c_int, [_vp, _vp]),
    "sgpr_get_neighbors": (C.c_int, [_vp, _vp, _vp, _vp]),
    "sgpr_get_local": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int]),
    "sgpr_get_cov": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "sgpr_get_dims": (C.c_int, [_vp, _vp]),
    "sgpr_profile": (C.c_int, [_vp, C.c_int]),
    "sgpr_get_stage_times": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int]),
}


def _preload_host_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as
    /opt/rocm's); whichever is mapped first serves both.  If the system runtime wins, a later
    `import torch` finds no GPU ("No HIP GPUs are available").  Hosts that use torch for device
    memory / torch.distributed next to this library therefore get torch's runtime mapped first —
    located via importlib, WITHOUT importing torch."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    # (librccl.so, which the library links for its all-reduce, is NOT preloaded: torch's copy mapped ahead
    # of torch's own load order ends the process with a double free at exit; whichever librccl.so.1 is
    # mapped first — torch's if torch was imported before, else /opt/rocm's — serves both)
    for name in ("libamdhip64.so",):
        cand = os.path.join(os.path.dirname(spec.origin), "lib", name)
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def load():
    """Load the shared library (no GPU needed to load; needed for every compute call)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with autoforce_amd/csrc/build.sh "
                "(or __graft_entry__.build()).  There is no CPU fallback."
            )
        _preload_host_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if os.environ.get("SGPR_API_TRACE"):
            lib = _Traced(lib)
        _lib = lib
    return _lib


class _Traced:
    """SGPR_API_TRACE=1: per-entry-point wall time, the device drained before and after each call (so
    asynchronous work is billed to the call that queued it); the table is printed at exit."""

    def __init__(self, lib):
        import atexit
        import time
        self._lib, self._time, self._acc = lib, time.perf_counter, {}
        try:
            self._sync = C.CDLL(None).hipDeviceSynchronize  # the runtime that is already mapped
        except AttributeError:
            self._sync = C.CDLL("libamdhip64.so.7").hipDeviceSynchronize
        atexit.register(self._report)

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in ("sgpr_last_error", "sgpr_version", "sgpr_device_count", "sgpr_packed_len", "sgpr_destroy"):
            return fn

        def call(*a):
            self._sync()
            t0 = self._time()
            r = fn(*a)
            t1 = self._time()
            self._sync()
            t2 = self._time()
            acc = self._acc.setdefault(name, [0, 0.0, 0.0])
            acc[0] += 1; acc[1] += t1 - t0; acc[2] += t2 - t1
            return r
        return call

    def _report(self):
        import sys
        print("# SGPR_API_TRACE: entry point, calls, ms in the call (mean), ms drained after it (mean), total ms", file=sys.stderr)
        for name, (n, a, b) in sorted(self._acc.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
            print(f"# {name:32s} {n:6d} {1e3 * a / n:10.3f} {1e3 * b / n:10.3f} {1e3 * (a + b):10.1f}", file=sys.stderr)


def check(code):
    if code != 0:
        raise SgprError(code, load().sgpr_last_error().decode())
    return code


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def device_count():
    return load().sgpr_device_count()
