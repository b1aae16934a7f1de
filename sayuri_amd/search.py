"""Python face of the search / self-play engine (libsayuri_host.so, csrc/engine/engine_capi.cc).

`SearchApi` binds either the product engine (`sayuri_engine_*`) or, in tests, the oracle taps on the
reference (`ref_*`, oracle/ref_search_driver.cc); both export the same result layouts.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .engine import Game

vp, ci, cf, u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint64

# Search::OptionTag
TAG_THINKING, TAG_FORCED, TAG_UNREUSED, TAG_NO_EXPLORING, TAG_NO_BUFFER = 2, 16, 32, 64, 128

RESULT_INTS = ("best_move", "best_no_pass_move", "random_move", "gumbel_move", "gumbel_no_pass_move",
               "capture_all_dead_move", "high_priority_move", "visits", "playouts", "to_move", "side_resign")
RESULT_FLOATS = ("root_eval", "root_score_lead", "best_eval", "root_score_stddev", "root_eval_stddev", "policy_kld")

_bound = False


def lib() -> ctypes.CDLL:
    global _bound
    h = _lib.host()
    if not _bound:
        h.sayuri_engine_net_new_callback.restype = vp
        h.sayuri_engine_net_new_callback.argtypes = [vp, ci, vp, ci, ctypes.c_char_p]
        h.sayuri_engine_net_new_pipe.restype = vp
        h.sayuri_engine_net_new_pipe.argtypes = [vp, ci, ctypes.c_char_p]
        h.sayuri_engine_net_free.argtypes = [vp]
        h.sayuri_engine_net_queries.restype = ctypes.c_ulong
        h.sayuri_engine_net_queries.argtypes = [vp]
        h.sayuri_engine_net_output.argtypes = [vp, vp, ci, ci, cf, ci, u64, vp]
        h.sayuri_engine_search_new.restype = vp
        h.sayuri_engine_search_new.argtypes = [vp, vp, ctypes.c_char_p]
        h.sayuri_engine_search_free.argtypes = [vp]
        h.sayuri_engine_search_seed.argtypes = [vp, u64, u64]
        h.sayuri_engine_search_computation.argtypes = [vp, vp, ci, ci] + [vp] * 6
        h.sayuri_engine_search_selfplay_move.restype = ci
        h.sayuri_engine_search_selfplay_move.argtypes = [vp, vp, ci]
        h.sayuri_engine_search_think.restype = ci
        h.sayuri_engine_search_think.argtypes = [vp, vp]
        h.sayuri_engine_search_update_territory_helper.argtypes = [vp]
        h.sayuri_engine_search_single_candidate.restype = ci
        h.sayuri_engine_search_single_candidate.argtypes = [vp, vp, ci]
        h.sayuri_engine_search_gather.restype = ctypes.c_long
        h.sayuri_engine_search_gather.argtypes = [vp, vp, ctypes.c_long]
        h.sayuri_selfplay_run.argtypes = [vp, ci, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_double, ci, vp, vp]
        h.sayuri_engine_last_error.restype = ctypes.c_char_p
        h.sayuri_pipe_raw.restype = vp
        h.sayuri_pipe_raw.argtypes = [vp]
        h.sayuri_pipe_weights_version.restype = ci
        h.sayuri_pipe_weights_version.argtypes = [vp]
        _bound = True
    return h


def options_text(opts: dict) -> bytes:
    """{"playouts": 400, "dirichlet_noise": True, "selfplay_query": ["bkp:19:7.5:1"]} -> b"playouts=400 ..."."""
    words = []
    for k, v in opts.items():
        for item in (v if isinstance(v, (list, tuple)) else [v]):
            if isinstance(item, bool):
                item = int(item)
            words.append(f"{k}={item}")
    return " ".join(words).encode()


class Network:
    """Evaluation facade.  backend: a HipForwardPipe (`pipe=`), a C forward function (`callback=`, tests),
    or nothing (the dummy random-output backend, like the reference without a weights file)."""

    def __init__(self, pipe=None, callback=None, callback_kind: int = 0, callback_user=None, weights_version: int = 4,
                 options: dict | None = None):
        h = lib()
        text = options_text(options or {})
        if pipe is not None:
            raw = h.sayuri_pipe_raw(pipe._h)
            self._h = h.sayuri_engine_net_new_pipe(raw, h.sayuri_pipe_weights_version(pipe._h), text)
            self._keep = pipe
        else:
            self._h = h.sayuri_engine_net_new_callback(callback, callback_kind, callback_user, weights_version, text)
            self._keep = (callback, callback_user)

    def close(self):
        if self._h:
            lib().sayuri_engine_net_free(self._h)
            self._h = None

    __del__ = close

    def queries(self) -> int:
        return int(lib().sayuri_engine_net_queries(self._h))

    def output(self, game: Game, ensemble: int = 0, symmetry: int = 0, temperature: float = 1.0, use_cache: bool = False,
               seed: int = 0) -> np.ndarray:
        out = np.zeros(2 * game.n + 9, np.float32)
        lib().sayuri_engine_net_output(self._h, game._h, ensemble, symmetry, temperature, int(use_cache), seed, out.ctypes.data)
        return out


class Search:
    def __init__(self, game: Game, network: Network, options: dict | None = None, seeds=(1, 2)):
        self._game, self._net = game, network
        self._h = lib().sayuri_engine_search_new(game._h, network._h, options_text(options or {}))
        lib().sayuri_engine_search_seed(self._h, seeds[0], seeds[1])

    def close(self):
        if self._h:
            lib().sayuri_engine_search_free(self._h)
            self._h = None

    __del__ = close

    def computation(self, playouts: int, tag: int = 0) -> dict:
        n = self._game.n
        ints, fl = np.zeros(16, np.int32), np.zeros(8, np.float32)
        vis, eq = np.zeros(n + 1, np.int32), np.zeros(n + 1, np.float32)
        tg, own = np.zeros(n + 1, np.float32), np.zeros(n, np.float32)
        lib().sayuri_engine_search_computation(self._h, self._game._h, playouts, tag, ints.ctypes.data, fl.ctypes.data,
                                               vis.ctypes.data, eq.ctypes.data, tg.ctypes.data, own.ctypes.data)
        out = {k: int(ints[i]) for i, k in enumerate(RESULT_INTS)}
        out.update({k: float(fl[i]) for i, k in enumerate(RESULT_FLOATS)})
        out.update(root_visits=vis, estimated_q=eq, target_policy=tg, ownership=own)
        return out

    def selfplay_move(self, tag: int = 0) -> int:
        return int(lib().sayuri_engine_search_selfplay_move(self._h, self._game._h, tag))

    def think(self) -> int:
        return int(lib().sayuri_engine_search_think(self._h, self._game._h))

    def single_candidate_records(self) -> list:
        """Indices of the buffered training samples whose search was cut short because the root had a single
        candidate move (see csrc/engine/search.h).  Call before gather_training_text()."""
        out = np.zeros(4096, np.int32)
        n = lib().sayuri_engine_search_single_candidate(self._h, out.ctypes.data, len(out))
        return out[:n].tolist()

    def update_territory_helper(self):
        lib().sayuri_engine_search_update_territory_helper(self._h)

    def gather_training_text(self) -> bytes:
        buf = ctypes.create_string_buffer(4 << 20)
        n = lib().sayuri_engine_search_gather(self._h, buf, len(buf))
        if n > len(buf):  # the text stays parked on the C side until a large enough buffer comes
            buf = ctypes.create_string_buffer(n)
            n = lib().sayuri_engine_search_gather(self._h, buf, len(buf))
        return buf.raw[:n]


STAT_NAMES = ("games_started", "games_done", "moves", "playouts", "nn_queries", "cache_lookups", "cache_hits", "records",
              "chunks_saved")


STATS_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_double, ctypes.c_int, ctypes.c_void_p)


def selfplay(pipe=None, options: dict | None = None, seconds: float = 0.0, move_cap: int = 0, name_suffix: str = "",
             weights_version: int = 4, on_stats=None, stats_interval: float = 2.0) -> dict:
    """Run the self-play loop (csrc/engine/selfplay.cc).  `pipe` = sayuri_amd.pipe.HipForwardPipe, or None for the
    dummy backend.  Returns the counters plus `elapsed` seconds.

    on_stats(stats: dict, local_halt: bool) -> bool is called every `stats_interval` seconds on the calling thread with a
    snapshot of the counters and this process's own halt wish (newer weights appeared in options["weights_dir"]); a
    true return winds the loop down (reference pipe.cc:246-258).  The multi-GPU driver all-gathers there
    (sayuri_amd.shard.PeriodicGather)."""
    h = lib()
    raw, version = None, weights_version
    if pipe is not None:
        raw = h.sayuri_pipe_raw(pipe._h)
        version = h.sayuri_pipe_weights_version(pipe._h)
    stats, el = np.zeros(20, np.uint64), ctypes.c_double(0)
    failure = []

    def hook(st, elapsed, local_halt, _user):
        try:
            snap = {k: int(st[i]) for i, k in enumerate(STAT_NAMES)}
            snap["finished_moves"], snap["prerolled_moves"] = int(st[10]), int(st[11])
            snap["chunks_saved_window"], snap["writer_cpu_seconds"] = int(st[12]), st[14] / 1e9
            snap["elapsed"] = float(elapsed)
            return 1 if on_stats(snap, bool(local_halt)) else 0
        except BaseException as e:  # an exception must not unwind through the C frames
            failure.append(e)
            return 1

    cb = STATS_FN(hook) if on_stats is not None else ctypes.cast(None, STATS_FN)
    h.sayuri_selfplay_run_ex.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_double, ctypes.c_int,
                                         STATS_FN, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    if h.sayuri_selfplay_run_ex(raw, version, options_text(options or {}), name_suffix.encode(), seconds, move_cap, cb, None,
                                float(stats_interval), stats.ctypes.data, ctypes.byref(el)):
        raise RuntimeError(h.sayuri_engine_last_error().decode())
    if failure:
        raise failure[0]
    out = {k: int(stats[i]) for i, k in enumerate(STAT_NAMES)}
    out["max_games"] = int(stats[9])
    out["finished_moves"], out["prerolled_moves"] = int(stats[10]), int(stats[11])
    # the data writer (csrc/engine/selfplay.cc WriterLoop): `_window` = when the time window ended, the others after the
    # final flush of its pool
    out["chunks_saved_window"] = int(stats[12])
    out["writer_cpu_seconds"], out["writer_cpu_seconds_window"] = stats[13] / 1e9, stats[14] / 1e9
    out["bytes_written"], out["text_bytes"], out["writer_flush_seconds"] = int(stats[15]), int(stats[16]), stats[17] / 1e9
    out["elapsed"] = el.value
    return out


def benchmark(pipe=None, options: dict | None = None, positions: int = 10, concurrent: int = 1, weights_version: int = 4) -> dict:
    """The reference's `--mode benchmark` (src/benchmark/benchmark.cc:110-161): policy-sampled openings, one timed
    Search::Computation each, playouts per second (csrc/engine/benchmark.cc)."""
    h = lib()
    raw, version = None, weights_version
    if pipe is not None:
        raw = h.sayuri_pipe_raw(pipe._h)
        version = h.sayuri_pipe_weights_version(pipe._h)
    out = np.zeros(8, np.float64)
    h.sayuri_engine_benchmark.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    if h.sayuri_engine_benchmark(raw, version, options_text(options or {}), positions, concurrent, out.ctypes.data):
        raise RuntimeError(h.sayuri_engine_last_error().decode())
    keys = ("playouts_per_move", "playouts_per_second_per_search", "playouts_per_second_total", "nn_evals_per_second",
            "wall_seconds", "elo", "nn_queries", "positions")
    return dict(zip(keys, (float(x) for x in out)))
