"""ctypes bindings of the two CPU checkers (TEST INFRASTRUCTURE ONLY).

* `PortNet`  -- oracle/libsayuri_oracle.so, our C restatement (travels to the GPU box).
* `RefNet`   -- oracle/_ref/libsayuri_ref.so, the reference's own sources compiled by
                oracle/Makefile (exists only where it was built; a process-wide singleton
                because the reference keeps global option state).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
PORT_SO = os.path.join(ORACLE_DIR, "libsayuri_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libsayuri_ref.so")

TAIL = 9  # pass, wdl[3], stm, score, q_err, score_err, offset


def build_port(force: bool = False) -> str:
    src = os.path.join(ORACLE_DIR, "sayuri_oracle.c")
    if force or not os.path.exists(PORT_SO) or os.path.getmtime(PORT_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "port"], stdout=subprocess.DEVNULL)
    return PORT_SO


def _fp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


class PortNet:
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            lib = ctypes.CDLL(build_port())
            lib.so_load.restype = ctypes.c_void_p
            lib.so_load.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
            lib.so_free.argtypes = [ctypes.c_void_p]
            lib.so_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
            lib.so_block_info.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
            lib.so_get_tensor.restype = ctypes.c_long
            lib.so_get_tensor.argtypes = [ctypes.c_void_p, ctypes.c_char_p,
                                          ctypes.POINTER(ctypes.c_float), ctypes.c_long]
            lib.so_forward.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
            lib.so_forward_raw.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.POINTER(ctypes.c_float)] * 5
            lib.so_postprocess.argtypes = [ctypes.c_int, ctypes.c_float,
                                           ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
            cls._lib = lib
        return cls._lib

    def __init__(self, path: str, winograd: bool = True):
        lib = self.lib()
        err = ctypes.create_string_buffer(256)
        self._h = lib.so_load(path.encode(), int(winograd), err, 256)
        if not self._h:
            raise RuntimeError(f"oracle loader: {err.value.decode()}")
        info = (ctypes.c_int * 12)()
        lib.so_info(self._h, info)
        self.info = list(info)

    def close(self):
        if self._h:
            self.lib().so_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def block_info(self, i: int):
        b = (ctypes.c_int * 5)()
        if self.lib().so_block_info(self._h, i, b):
            raise IndexError(i)
        return list(b)

    def tensor(self, name: str) -> Optional[np.ndarray]:
        n = self.lib().so_get_tensor(self._h, name.encode(), None, 0)
        if n < 0:
            return None
        out = np.zeros(n, np.float32)
        if n:
            self.lib().so_get_tensor(self._h, name.encode(), _fp(out), n)
        return out

    def forward(self, planes: np.ndarray, board_size: int, komi: float = 7.5, offset: int = 0) -> np.ndarray:
        planes = np.ascontiguousarray(planes, np.float32)
        out = np.zeros(2 * board_size * board_size + TAIL, np.float32)
        if self.lib().so_forward(self._h, board_size, komi, offset, _fp(planes), _fp(out)):
            raise RuntimeError("so_forward failed")
        return out

    def forward_raw(self, planes: np.ndarray, board_size: int):
        """-> prob [prob_ch][S], pass [pass_outs], misc [misc_outs], own [S]"""
        planes = np.ascontiguousarray(planes, np.float32)
        s = board_size * board_size
        prob = np.zeros((self.info[6], s), np.float32)
        pas = np.zeros(self.info[7], np.float32)
        misc = np.zeros(self.info[9], np.float32)
        own = np.zeros((self.info[8], s), np.float32)
        if self.lib().so_forward_raw(self._h, board_size, _fp(planes), _fp(prob), _fp(pas), _fp(misc), _fp(own)):
            raise RuntimeError("so_forward_raw failed")
        return prob, pas, misc, own[0]

    @classmethod
    def postprocess(cls, raw: np.ndarray, board_size: int, temp: float = 1.0) -> np.ndarray:
        raw = np.ascontiguousarray(raw, np.float32)
        post = np.zeros(2 * board_size * board_size + 1 + 8, np.float32)
        cls.lib().so_postprocess(board_size, temp, _fp(raw), _fp(post))
        return post


def ref_available() -> bool:
    return os.path.exists(REF_SO)


class RefNet:
    """The reference's own loader + BlasForwardPipe.  One network at a time per process."""
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            lib = ctypes.CDLL(REF_SO)
            lib.ref_last_error.restype = ctypes.c_char_p
            lib.ref_init.argtypes = [ctypes.c_char_p, ctypes.c_int]
            lib.ref_get_tensor.restype = ctypes.c_long
            lib.ref_get_tensor.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_float), ctypes.c_long]
            lib.ref_forward.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
            cls._lib = lib
        return cls._lib

    def __init__(self, path: str, winograd: bool = True):
        lib = self.lib()
        if lib.ref_init(path.encode(), int(winograd)):
            raise RuntimeError(f"reference loader: {lib.ref_last_error().decode()}")
        info = (ctypes.c_int * 12)()
        lib.ref_info(info)
        self.info = list(info)

    def block_info(self, i: int):
        b = (ctypes.c_int * 5)()
        if self.lib().ref_block_info(i, b):
            raise IndexError(i)
        return list(b)

    def tensor(self, name: str) -> Optional[np.ndarray]:
        n = self.lib().ref_get_tensor(name.encode(), None, 0)
        if n < 0:
            return None
        out = np.zeros(n, np.float32)
        if n:
            self.lib().ref_get_tensor(name.encode(), _fp(out), n)
        return out

    def forward(self, planes: np.ndarray, board_size: int, komi: float = 7.5, offset: int = 0) -> np.ndarray:
        planes = np.ascontiguousarray(planes, np.float32)
        out = np.zeros(2 * board_size * board_size + TAIL, np.float32)
        if self.lib().ref_forward(board_size, komi, 0, offset, _fp(planes), _fp(out)):
            raise RuntimeError(f"ref_forward: {self.lib().ref_last_error().decode()}")
        return out
