"""mvolps_amd -- MI355X-native dense-simplex LP-relaxation engine behind MVOLPS's node solve.

The product is libmvolps_amd.so (C ABI in include/mvx.h, hand-written gfx950 kernels).
This package is the thin Python binding used by the tests, bench.py and the multi-GPU
branch-and-bound coordinator.  There is no CPU fallback: engine calls need a HIP device.
"""
import ctypes as C
import os

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmvolps_amd.so")

_EXTRA = {
    "device_count": (C.c_int, []),
    "set_device": (C.c_int, [C.c_int]),
    "profile_enable": (None, [C.c_int]),
    "profile_reset": (None, []),
    "profile_update_ms": (C.c_double, []),
    "profile_update_launches": (C.c_longlong, []),
    "last_solve_ms": (C.c_double, [C.c_void_p]),
    "sync": (None, []),
    "set_tuning": (None, [C.c_int, C.c_int, C.c_int]),
    "set_batch_slots": (None, [C.c_int]),
    "set_persist": (None, [C.c_int]),
    "set_chain": (None, [C.c_int]),
    "set_cluster": (None, [C.c_int]),
    "cluster_stats": (None, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "set_dual_chain": (None, [C.c_int]),
    "set_refresh": (None, [C.c_int, C.c_double]),
    "get_refresh_cnt": (C.c_int, [C.c_void_p]),
    "row_residual": (C.c_double, [C.c_void_p]),
    "persist_stats": (None, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "last_error": (C.c_int, []),
    "bind_thread": (C.c_int, []),
    "persist_cycles": (None, [C.POINTER(C.c_ulonglong)]),
    "simplex_batch": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "pack_size": (C.c_longlong, [C.c_void_p]),
    "pack": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pack_size_from": (C.c_longlong, [C.c_void_p, C.c_void_p]),
    "pack_from": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
}

_api = None


def load_library():
    """dlopen the engine; raises if it has not been built (python -m mvolps_amd.build)."""
    # torch bundles its own HIP runtime; whichever libamdhip64 is loaded first serves the whole
    # process, and torch cannot find the GPU when it is not its own.  Load torch's first so that
    # the engine, torch.cuda and torch.distributed (RCCL) share one runtime.
    if os.environ.get("MVX_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libmvolps_amd.so is missing (%s): build it with `python -m mvolps_amd.build`; "
            "there is no fallback path" % LIB_PATH
        )
    return C.CDLL(LIB_PATH)


def api():
    """Function table of the HIP engine (prefix mvx_)."""
    global _api
    if _api is None:
        _api = capi.LpApi(load_library(), "mvx_", _EXTRA)
    return _api


def require_device():
    a = api()
    n = a.device_count()
    if n <= 0:
        raise RuntimeError("no HIP device visible: the gfx950 engine cannot run and there is no CPU fallback")
    return n


def create_prob():
    require_device()
    return api().create()
