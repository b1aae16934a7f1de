"""Python face of the native Go engine (board, game state, network input encoder).

A thin ctypes wrapper over the `sayuri_go_*` entry points of libsayuri_host.so
(sayuri_amd/csrc/engine/go_capi.cc).  Moves are intersection indices: 0..N-1 row-major from the
first row, N = pass, -1 = resign.  Colours: 0 black, 1 white, 2 empty.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

BLACK, WHITE, EMPTY, WALL = 0, 1, 2, 3
AREA, TERRITORY = 0, 1
MAP_NAMES = ("cell", "legal", "liberties", "ladder", "safe_area", "ownership", "raw_ownership", "seki", "tactics")
INFO_NAMES = ("hash", "ko_hash", "to_move", "last_move", "ko_move", "passes", "prisoners_black", "prisoners_white",
              "move_number", "superko", "game_over", "handicap", "winner", "board_size", "scoring", "symmetry_hashes")
SCALAR_NAMES = ("komi", "komi_with_penalty", "wave", "final_score_black", "penalty", "penalty_offset")


class GoApi:
    """Binds one library that exports the `<prefix>new/play/...` family (the product engine, or the
    reference tap of oracle/ref_game_driver.cc in tests)."""

    def __init__(self, lib: ctypes.CDLL, prefix: str):
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float

        def fn(name, restype, *argtypes):
            f = getattr(lib, prefix + name)
            f.restype, f.argtypes = restype, list(argtypes)
            return f

        self.new = fn("new", vp, ci, cf, ci)
        self.clone = fn("clone", vp, vp)
        self.free = fn("free", None, vp)
        self.play = fn("play", ci, vp, ci, ci)
        self.append = fn("append", ci, vp, ci, ci)
        self.undo = fn("undo", ci, vp)
        self.fixed_handicap = fn("fixed_handicap", ci, vp, ci)
        self.set_komi = fn("set_komi", None, vp, cf)
        self.set_rule = fn("set_rule", None, vp, ci)
        self.set_to_move = fn("set_to_move", None, vp, ci)
        self.info = fn("info", None, vp, vp)
        self.scalars = fn("scalars", None, vp, vp)
        self.planes = fn("planes", ci, vp, ci, ci, vp)
        # product engine only (the reference tap has no compact encoder)
        self.planes_packed = fn("planes_packed", ci, vp, ci, ci, vp) if hasattr(lib, prefix + "planes_packed") else None
        self.maps = fn("maps", None, vp, vp)
        self.set_helper = fn("set_territory_helper_from_ownership", None, vp)


_api = None


def api() -> GoApi:
    global _api
    if _api is None:
        _api = GoApi(_lib.host(), "sayuri_go_")
    return _api


class Game:
    """One game state.  `api` defaults to the product engine."""

    def __init__(self, board_size: int = 19, komi: float = 7.5, scoring: int = AREA, api_: GoApi | None = None,
                 _handle=None):
        self._a = api_ or api()
        self.board_size = board_size
        self.n = board_size * board_size
        self._h = _handle if _handle is not None else self._a.new(board_size, komi, scoring)

    def close(self):
        if self._h:
            self._a.free(self._h)
            self._h = None

    __del__ = close

    def clone(self) -> "Game":
        return Game(self.board_size, api_=self._a, _handle=self._a.clone(self._h))

    def play(self, move: int, color: int = -1) -> bool:
        return bool(self._a.play(self._h, move, color))

    def append(self, move: int, color: int) -> bool:
        return bool(self._a.append(self._h, move, color))

    def undo(self) -> bool:
        return bool(self._a.undo(self._h))

    def fixed_handicap(self, n: int) -> bool:
        return bool(self._a.fixed_handicap(self._h, n))

    def set_komi(self, komi: float):
        self._a.set_komi(self._h, komi)

    def set_rule(self, scoring: int):
        self._a.set_rule(self._h, scoring)

    def set_to_move(self, color: int):
        self._a.set_to_move(self._h, color)

    def set_territory_helper_from_ownership(self):
        self._a.set_helper(self._h)

    def info(self) -> np.ndarray:
        out = np.zeros(16, np.uint64)
        self._a.info(self._h, out.ctypes.data)
        return out

    def scalars(self) -> np.ndarray:
        out = np.zeros(6, np.float32)
        self._a.scalars(self._h, out.ctypes.data)
        return out

    def maps(self) -> np.ndarray:
        out = np.zeros((9, self.n + 1), np.uint8)
        self._a.maps(self._h, out.ctypes.data)
        return out

    def planes_packed(self, symmetry: int = 0, weights_version: int = 4):
        """The compact encoder (csrc/host/packed_planes.h): (record uint32[binary*12 + 8], binary plane count)."""
        rec = np.zeros(40 * 12 + 8, np.uint32)
        binary = self._a.planes_packed(self._h, symmetry, weights_version, rec.ctypes.data)
        return rec[:binary * 12 + 8].copy(), binary

    def planes(self, symmetry: int = 0, weights_version: int = 4) -> np.ndarray:
        channels = 38 if weights_version in (1, 2) else 43
        out = np.zeros((channels, self.n), np.float32)
        self._a.planes(self._h, symmetry, weights_version, out.ctypes.data)
        return out


PACKED_WORDS = 12    # uint32 words per bit plane (19 x 19 cells)
PACKED_SCALARS = 8   # float slots behind the bit planes


def expand_packed(record: np.ndarray, binary: int, board_size: int, channels: int = 43) -> np.ndarray:
    """fp32 planes [channels][board_size^2] of a packed record (PackedPlanes::Expand, csrc/host/packed_planes.h)."""
    n = board_size * board_size
    rec = np.ascontiguousarray(record, np.uint32)
    bits = rec[:binary * PACKED_WORDS].reshape(binary, PACKED_WORDS)
    scal = rec[binary * PACKED_WORDS:binary * PACKED_WORDS + PACKED_SCALARS].view(np.float32)
    idx = np.arange(n)
    out = np.empty((channels, n), np.float32)
    out[:binary] = (bits[:, idx >> 5] >> (idx & 31).astype(np.uint32)) & 1
    for c in range(binary, channels):
        out[c] = scal[c - binary]
    return out


def pack_planes(planes: np.ndarray, binary: int = 37) -> np.ndarray:
    """Packed record of fp32 planes [channels][n] whose first `binary` planes are 0/1 and whose other planes are
    constant over the board (the shape of every encoder output and of the synthetic bench planes)."""
    planes = np.asarray(planes, np.float32)
    channels, n = planes.shape
    head = planes[:binary]
    if not np.all((head == 0) | (head == 1)) or not np.all(planes[binary:] == planes[binary:, :1]):
        raise ValueError("planes are not packable (binary planes must be 0/1, the others constant)")
    rec = np.zeros(binary * PACKED_WORDS + PACKED_SCALARS, np.uint32)
    bits = rec[:binary * PACKED_WORDS].reshape(binary, PACKED_WORDS)
    idx = np.arange(n)
    for c in range(binary):
        on = idx[head[c] != 0]
        np.bitwise_or.at(bits[c], on >> 5, (np.uint32(1) << (on & 31).astype(np.uint32)))
    rec[binary * PACKED_WORDS:binary * PACKED_WORDS + channels - binary] = planes[binary:, 0].view(np.uint32)
    return rec
