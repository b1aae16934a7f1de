"""ctypes bindings of the native libraries.  Loading fails LOUDLY: there is no CPU fallback
for the product path (the CPU oracle lives under oracle/ and is test infrastructure)."""
from __future__ import annotations

import ctypes
import os

from . import _build

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)


class KernelStat(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("launches", ctypes.c_int32), ("total_ms", ctypes.c_float),
                ("flops", ctypes.c_double), ("bytes", ctypes.c_double)]


# every symbol include/sayuri_hip.h declares
HIP_SYMBOLS = [
    "sayuri_hip_device_count", "sayuri_hip_create", "sayuri_hip_load_tensor", "sayuri_hip_forward",
    "sayuri_hip_submit", "sayuri_hip_wait", "sayuri_hip_query", "sayuri_hip_upload", "sayuri_hip_run", "sayuri_hip_sync", "sayuri_hip_download", "sayuri_hip_time_runs",
    "sayuri_hip_forward_packed", "sayuri_hip_submit_packed", "sayuri_hip_profile_run", "sayuri_hip_mark_kernel", "sayuri_hip_timed_stat", "sayuri_hip_host_alloc", "sayuri_hip_host_free", "sayuri_hip_device_bytes", "sayuri_hip_last_chains", "sayuri_hip_tower_state",
    "sayuri_hip_destroy", "sayuri_hip_last_error", "sayuri_hip_test_conv", "sayuri_hip_test_last_conv_kind",
    "sayuri_hip_test_se_unit", "sayuri_hip_test_head_tail", "sayuri_hip_test_conv_se", "sayuri_hip_test_head_board",
]

_hip = None
_host = None


def _require(path: str) -> str:
    if not os.path.exists(path):
        raise RuntimeError(
            f"native library {path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the MI355X forward pipe)")
    return path


def hip() -> ctypes.CDLL:
    global _hip
    if _hip is None:
        path = _build.HIP_SO
        fake = os.environ.get("SAYURI_FAKE_HIP_LIB")
        if fake:
            # TEST HOOK (tests/fake_hip): a CPU stand-in for the DEVICE side of include/sayuri_hip.h, loaded ahead of the real
            # library so that the host side can be run without a GPU.  Never set in production; said out loud when it is.
            import sys
            print(f"[sayuri_amd] SAYURI_FAKE_HIP_LIB is set: the device side is the test stand-in {fake}, NOT the MI355X engine",
                  file=sys.stderr)
            path = fake
        lib = ctypes.CDLL(_require(path), mode=ctypes.RTLD_GLOBAL)
        lib.sayuri_hip_last_error.restype = ctypes.c_char_p
        lib.sayuri_hip_device_count.restype = ctypes.c_int
        lib.sayuri_hip_upload.argtypes = [ctypes.c_void_p, ctypes.c_int, c_float_p, c_int_p]
        lib.sayuri_hip_run.argtypes = [ctypes.c_void_p]
        lib.sayuri_hip_sync.argtypes = [ctypes.c_void_p]
        lib.sayuri_hip_download.argtypes = [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p]
        lib.sayuri_hip_forward.argtypes = [ctypes.c_void_p, ctypes.c_int, c_float_p, c_int_p, c_float_p,
                                           c_float_p, c_float_p, c_float_p]
        lib.sayuri_hip_time_runs.argtypes = [ctypes.c_void_p, ctypes.c_int, c_float_p]
        lib.sayuri_hip_profile_run.argtypes = [ctypes.c_void_p, ctypes.POINTER(KernelStat), ctypes.c_int]
        lib.sayuri_hip_mark_kernel.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        lib.sayuri_hip_timed_stat.argtypes = [ctypes.c_void_p, ctypes.POINTER(KernelStat)]
        lib.sayuri_hip_device_bytes.restype = ctypes.c_size_t
        lib.sayuri_hip_device_bytes.argtypes = [ctypes.c_void_p]
        lib.sayuri_hip_last_chains.restype = ctypes.c_int
        lib.sayuri_hip_last_chains.argtypes = [ctypes.c_void_p]
        lib.sayuri_hip_tower_state.restype = ctypes.c_int
        lib.sayuri_hip_tower_state.argtypes = [ctypes.c_void_p]
        if not fake:  # the stand-in has no kernels to tap
            lib.sayuri_hip_test_conv.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_int_p, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p]
        _hip = lib
    return _hip


def host() -> ctypes.CDLL:
    global _host
    if _host is None:
        hip()
        lib = ctypes.CDLL(_require(_build.HOST_SO))
        lib.sayuri_host_last_error.restype = ctypes.c_char_p
        lib.sayuri_weights_load.restype = ctypes.c_void_p
        lib.sayuri_weights_load.argtypes = [ctypes.c_char_p]
        lib.sayuri_weights_free.argtypes = [ctypes.c_void_p]
        lib.sayuri_weights_info.argtypes = [ctypes.c_void_p, c_int_p]
        lib.sayuri_weights_block_info.argtypes = [ctypes.c_void_p, ctypes.c_int, c_int_p]
        lib.sayuri_weights_tensor.restype = ctypes.c_long
        lib.sayuri_weights_tensor.argtypes = [ctypes.c_void_p, ctypes.c_char_p, c_float_p, ctypes.c_long]
        lib.sayuri_pipe_create.restype = ctypes.c_void_p
        lib.sayuri_pipe_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int]
        lib.sayuri_pipe_destroy.argtypes = [ctypes.c_void_p]
        lib.sayuri_pipe_num_workers.argtypes = [ctypes.c_void_p]
        lib.sayuri_pipe_ctx.restype = ctypes.c_void_p
        lib.sayuri_pipe_ctx.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.sayuri_pipe_reconstruct.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        lib.sayuri_pipe_eval.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p,
                                         c_int_p, c_float_p, c_int_p, c_float_p]
        lib.sayuri_pipe_pump_times.restype = None
        lib.sayuri_pipe_pump_times.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                               ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]
        lib.sayuri_pipe_netbench.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                             ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]
        _host = lib
    return _host
