"""Python face of the host library: the weights loader and HipForwardPipe, with the method
names of the reference's plugin interface (src/neural/network_basic.h:132-161)."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import numpy as np

from . import _lib

PLANES_LEN = 43 * 361  # InputData::planes
OUT_LEN = 2 * 361 + 9
MAX_BOARD = 19


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_lib.c_float_p)


def _ip(a: np.ndarray):
    return a.ctypes.data_as(_lib.c_int_p)


class Weights:
    """DNNWeights as parsed + BN-folded by the product loader (csrc/host/weights_loader.cc)."""

    def __init__(self, path: str):
        lib = _lib.host()
        self._h = lib.sayuri_weights_load(path.encode())
        if not self._h:
            raise RuntimeError(f"Fail to load the network file! Cause: {lib.sayuri_host_last_error().decode()}")
        info = (ctypes.c_int * 12)()
        lib.sayuri_weights_info(self._h, info)
        self.info = list(info)

    def block_info(self, i: int) -> List[int]:
        b = (ctypes.c_int * 5)()
        if _lib.host().sayuri_weights_block_info(self._h, i, b):
            raise IndexError(i)
        return list(b)

    def tensor(self, name: str) -> Optional[np.ndarray]:
        lib = _lib.host()
        n = lib.sayuri_weights_tensor(self._h, name.encode(), None, 0)
        if n < 0:
            return None
        out = np.zeros(n, np.float32)
        if n:
            lib.sayuri_weights_tensor(self._h, name.encode(), _fp(out), n)
        return out

    def close(self):
        if self._h:
            _lib.host().sayuri_weights_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipForwardPipe:
    """One process, one pipe; `device` = -1 uses every visible GPU (one pump thread each)."""

    def __init__(self, weights_path: str, board_size: int = MAX_BOARD, batch_size: int = 256, fp16: bool = True,
                 device: int = 0, waittime_ms: int = 2):
        lib = _lib.host()
        self._h = lib.sayuri_pipe_create(weights_path.encode(), board_size, batch_size, int(fp16), device,
                                         waittime_ms)
        if not self._h:
            raise RuntimeError(f"HipForwardPipe: {lib.sayuri_host_last_error().decode()}")
        self.board_size = board_size
        self.batch_size = batch_size
        self.fp16 = fp16

    # -- NetworkForwardPipe surface
    def Valid(self) -> bool:
        return self._h is not None

    def GetNumWorkers(self) -> int:
        return _lib.host().sayuri_pipe_num_workers(self._h)

    def Construct(self, board_size: int, batch_size: int):
        if _lib.host().sayuri_pipe_reconstruct(self._h, board_size, batch_size):
            raise RuntimeError(_lib.host().sayuri_host_last_error().decode())
        self.board_size, self.batch_size = board_size, batch_size

    def Destroy(self):
        if self._h:
            _lib.host().sayuri_pipe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.Destroy()
        except Exception:
            pass

    def netbench(self, threads: int, seconds: float = 5.0, board: int = MAX_BOARD):
        """The reference's `netbench` measure (gtp.cc:1468-1568): threads blocked in Forward()
        for `seconds`; returns (evals_per_sec, total_evals).  Includes queue + PCIe both ways."""
        eps, tot = ctypes.c_double(0), ctypes.c_long(0)
        if _lib.host().sayuri_pipe_netbench(self._h, threads, seconds, board, ctypes.byref(eps), ctypes.byref(tot)):
            raise RuntimeError(_lib.host().sayuri_host_last_error().decode())
        return eps.value, tot.value

    def pump_times(self):
        """-> dict of pump-thread time (us) since construction + batch / eval counters."""
        t = (ctypes.c_double * 8)()
        b, e = ctypes.c_long(0), ctypes.c_long(0)
        _lib.host().sayuri_pipe_pump_times(self._h, t, ctypes.byref(b), ctypes.byref(e))
        return {"forward_us": t[0], "fill_us": t[1], "wait_batch_us": t[2], "wait_copies_us": t[3],
                "wake_parked_us": t[4], "partial_batches": int(round(t[5])), "gpu_queue_empty_us": t[6],
                "wait_plane_copies_us": t[7], "batches": b.value, "evals": e.value}

    def ctx(self, gpu: int = 0) -> int:
        c = _lib.host().sayuri_pipe_ctx(self._h, gpu)
        if not c:
            raise RuntimeError("no such gpu in this pipe")
        return c

    def _eval(self, mode: int, planes: Sequence[np.ndarray], board_sizes: Sequence[int], komi=None, offsets=None,
              gpu: int = 0) -> List[np.ndarray]:
        n = len(planes)
        buf = np.zeros((n, PLANES_LEN), np.float32)
        for i, p in enumerate(planes):
            flat = np.ascontiguousarray(p, np.float32).ravel()
            buf[i, :flat.size] = flat
        bsz = np.asarray(board_sizes, np.int32)
        km = np.asarray(komi if komi is not None else [7.5] * n, np.float32)
        off = np.asarray(offsets if offsets is not None else [0] * n, np.int32)
        out = np.zeros((n, OUT_LEN), np.float32)
        if _lib.host().sayuri_pipe_eval(self._h, mode, gpu, n, _fp(buf), _ip(bsz), _fp(km), _ip(off), _fp(out)):
            raise RuntimeError(_lib.host().sayuri_host_last_error().decode())
        res = []
        for i in range(n):
            s = int(bsz[i]) ** 2
            res.append(np.concatenate([out[i, :s], out[i, 361:361 + s], out[i, 722:]]))
        return res

    def BatchForward(self, planes, board_sizes, komi=None, offsets=None, gpu: int = 0):
        """-> per sample: prob[bs*bs], own[bs*bs], pass, wdl[3], stm, score, q_err, score_err, offset
        (the packing of oracle so_forward / ref_forward)."""
        return self._eval(0, planes, board_sizes, komi, offsets, gpu)

    def Forward(self, planes, board_sizes, komi=None, offsets=None):
        """n concurrent blocking Forward() calls through the batching queue."""
        return self._eval(1, planes, board_sizes, komi, offsets)

    def ForwardPacked(self, planes, board_sizes, komi=None, offsets=None, mixed: bool = False):
        """The same through ForwardPacked() (csrc/host/packed_planes.h): the planes are packed into bit planes + scalars on
        the way in (they must be packable: 0/1 binary planes, constant scalar planes).  mixed: odd requests packed, even
        ones fp32, so that batches hold both kinds."""
        return self._eval(3 if mixed else 2, planes, board_sizes, komi, offsets)


def hip_forward_raw(ctx: int, planes_grid: np.ndarray, board_sizes, board: int, prob_ch: int = 5, pass_outs: int = 5,
                    misc_outs: int = 15):
    """sayuri_hip_forward on NN-grid planes [n][43][board*board] -> prob, pass, misc, own."""
    lib = _lib.hip()
    n = planes_grid.shape[0]
    planes_grid = np.ascontiguousarray(planes_grid, np.float32)
    bsz = np.asarray(board_sizes, np.int32)
    prob = np.zeros((n, prob_ch, board * board), np.float32)
    pas = np.zeros((n, pass_outs), np.float32)
    misc = np.zeros((n, misc_outs), np.float32)
    own = np.zeros((n, board * board), np.float32)
    if lib.sayuri_hip_forward(ctx, n, _fp(planes_grid), _ip(bsz), _fp(prob), _fp(pas), _fp(misc), _fp(own)):
        raise RuntimeError(lib.sayuri_hip_last_error().decode())
    return prob, pas, misc, own


def hip_forward_packed_raw(ctx: int, records: np.ndarray, binary: int, board_sizes, board: int, prob_ch: int = 5, pass_outs: int = 5,
                           misc_outs: int = 15):
    """sayuri_hip_forward_packed on packed records [n][binary*12 + 8] (uint32) -> prob, pass, misc, own."""
    lib = _lib.hip()
    lib.sayuri_hip_forward_packed.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, _lib.c_int_p, _lib.c_float_p,
                                              _lib.c_float_p, _lib.c_float_p, _lib.c_float_p]
    records = np.ascontiguousarray(records, np.uint32)
    n = records.shape[0]
    assert records.shape[1] == binary * 12 + 8
    bsz = np.asarray(board_sizes, np.int32)
    prob = np.zeros((n, prob_ch, board * board), np.float32)
    pas = np.zeros((n, pass_outs), np.float32)
    misc = np.zeros((n, misc_outs), np.float32)
    own = np.zeros((n, board * board), np.float32)
    if lib.sayuri_hip_forward_packed(ctx, n, records.ctypes.data, binary, _ip(bsz), _fp(prob), _fp(pas), _fp(misc), _fp(own)):
        raise RuntimeError(lib.sayuri_hip_last_error().decode())
    return prob, pas, misc, own
