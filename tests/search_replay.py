"""Shared by tests/golden/make_golden_search.py and tests/test_search_cpu.py: search configurations, the
reference-tap bindings and the record comparison."""
from __future__ import annotations

import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libsayuri_ref.so")

# option name -> C type of the reference's option map entry
INT_OPTS = {"playouts", "gumbel_prom_visits", "gumbel_considered_moves", "gumbel_playouts_threshold", "random_min_visits",
            "resign_playouts", "fastsearch_playouts", "kldgain_interval", "cache_memory_mib"}
BOOL_OPTS = {"dirichlet_noise", "gumbel", "first_pass_bonus", "early_symm_cache", "no_cache", "always_completed_q_policy",
             "symm_pruning", "reuse_tree", "capture_all_dead", "friendly_pass", "cpuct_dynamic", "use_stm_winrate"}
DOUBLE_OPTS = {"kldgain_per_node"}

# every option a configuration may touch, at the reference's default (so one run cannot leak into the next)
DEFAULTS = dict(playouts=400, dirichlet_noise=0, gumbel=0, first_pass_bonus=0, early_symm_cache=0, no_cache=0,
                random_moves_factor=0.0, root_policy_temp=1.0, policy_temp=1.0, lcb_reduction=0.02, resign_threshold=0.1,
                fastsearch_playouts=0, fastsearch_playouts_prob=0.0, resign_playouts=0, resign_discard_prob=0.0,
                random_fastsearch_prob=0.0, always_completed_q_policy=0, reuse_tree=0, symm_pruning=0, capture_all_dead=0,
                kldgain_interval=0, kldgain_per_node=0.0, forced_playouts_k=0.0, gumbel_playouts_threshold=400,
                gumbel_considered_moves=16, gumbel_prom_visits=1, cache_memory_mib=400, score_utility_factor=0.4)

# Fixed-seed self-play games on the DUMMY backend (random network outputs drawn from the seeded streams: the
# whole game is a pure function of the seed).  (seed, board, komi, scoring, options)
DUMMY_GAMES = [
    (1, 9, 7.0, 0, dict(playouts=200)),
    (2, 9, 7.0, 0, dict(playouts=300, dirichlet_noise=1, first_pass_bonus=1, random_moves_factor=0.1, early_symm_cache=1)),
    (3, 7, 9.0, 0, dict(playouts=400, gumbel=1, gumbel_playouts_threshold=40)),
    (4, 9, 6.0, 1, dict(playouts=200, first_pass_bonus=1)),
    (5, 13, 7.5, 0, dict(playouts=160, dirichlet_noise=1, fastsearch_playouts=60, fastsearch_playouts_prob=0.5, resign_playouts=30,
                         resign_discard_prob=0.5, random_fastsearch_prob=0.2, resign_threshold=0.2)),
    (6, 9, 7.0, 0, dict(playouts=200, reuse_tree=1, dirichlet_noise=1)),
    (7, 19, 7.5, 0, dict(playouts=100, dirichlet_noise=1, first_pass_bonus=1, symm_pruning=1, capture_all_dead=1)),
    (8, 9, 5.5, 1, dict(playouts=240, gumbel=1, always_completed_q_policy=1, first_pass_bonus=1)),
]

# Fixed-seed searches with a real network (synthetic 6b96 weights, seed 21): (seed, board, komi, scoring, options, moves)
NN_GAMES = [
    (11, 9, 7.0, 0, dict(playouts=24, dirichlet_noise=1, first_pass_bonus=1), 10),
    (12, 9, 7.0, 0, dict(playouts=24, gumbel=1, gumbel_playouts_threshold=16), 8),
    (13, 13, 6.5, 0, dict(playouts=16, early_symm_cache=1), 6),
    # BASELINE.json configs[0] at its stated size: 9x9, 6b x 96 net, the CPU pipe, one thread, 100 visits per move
    (14, 9, 7.0, 0, dict(playouts=100), 10),
]


# Fixed-seed games played with ThinkBestMove (the genmove path: resign / friendly-pass / capture-all-dead policy) on the
# dummy backend: (seed, board, komi, scoring, options)
THINK_GAMES = [
    (21, 9, 7.0, 0, dict(playouts=120, friendly_pass=1, capture_all_dead=1, resign_threshold=0.3)),
    (22, 9, 7.0, 0, dict(playouts=100, reuse_tree=1, friendly_pass=1, resign_threshold=0.25, random_moves_factor=0.1)),
    (23, 7, 9.0, 1, dict(playouts=150, resign_threshold=0.0)),
]


def options(extra: dict) -> dict:
    o = dict(DEFAULTS)
    o.update(extra)
    return o


class RefSearchApi:
    """ctypes bindings of oracle/ref_search_driver.cc."""

    def __init__(self):
        vp, ci, cf, u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint64
        lib = self.lib = ctypes.CDLL(REF_SO)
        lib.ref_net_new.restype = vp
        lib.ref_net_new.argtypes = [ctypes.c_char_p]
        lib.ref_net_free.argtypes = [vp]
        lib.ref_net_output.argtypes = [vp, vp, ci, ci, cf, ci, vp]
        lib.ref_search_new.restype = vp
        lib.ref_search_new.argtypes = [vp, vp]
        lib.ref_search_free.argtypes = [vp]
        lib.ref_seed.argtypes = [u64, u64]
        lib.ref_search_computation.argtypes = [vp, vp, ci, ci] + [vp] * 6
        lib.ref_search_selfplay_move.restype = ci
        lib.ref_search_selfplay_move.argtypes = [vp, vp, ci]
        lib.ref_search_think.restype = ci
        lib.ref_search_think.argtypes = [vp, vp]
        lib.ref_search_gather.restype = ctypes.c_long
        lib.ref_search_gather.argtypes = [vp, vp, ctypes.c_long]
        lib.ref_search_update_territory_helper.argtypes = [vp]
        lib.ref_opt_int.argtypes = [ctypes.c_char_p, ci]
        lib.ref_opt_float.argtypes = [ctypes.c_char_p, cf]
        lib.ref_opt_bool.argtypes = [ctypes.c_char_p, ci]
        lib.ref_opt_double.argtypes = [ctypes.c_char_p, ctypes.c_double]
        lib.ref_init.argtypes = [ctypes.c_char_p, ci]

    def set_options(self, opts: dict):
        for k, v in opts.items():
            kb = k.encode()
            if k in INT_OPTS:
                r = self.lib.ref_opt_int(kb, int(v))
            elif k in BOOL_OPTS:
                r = self.lib.ref_opt_bool(kb, int(v))
            elif k in DOUBLE_OPTS:
                r = self.lib.ref_opt_double(kb, float(v))
            else:
                r = self.lib.ref_opt_float(kb, float(v))
            assert r == 0, k


def ref_selfplay_game(api: RefSearchApi, go_api, seed, board, komi, scoring, opts, weights: bytes = b"", max_moves=100000):
    """Play one fixed-seed self-play game with the REFERENCE search; returns (moves, training-record text)."""
    from sayuri_amd.engine import Game
    api.set_options(options(opts))
    net = api.lib.ref_net_new(weights)
    game = Game(board, komi, scoring, api_=go_api)
    search = api.lib.ref_search_new(game._h, net)
    api.lib.ref_seed(seed, seed + 77)
    moves = []
    while not game.info()[10] and len(moves) < max_moves:
        mv = api.lib.ref_search_selfplay_move(search, game._h, 0)
        moves.append(mv)
        assert game.play(mv)
    api.lib.ref_search_update_territory_helper(search)
    buf = ctypes.create_string_buffer(64 << 20)
    n = api.lib.ref_search_gather(search, buf, len(buf))
    api.lib.ref_search_free(search)
    api.lib.ref_net_free(net)
    return moves, buf.raw[:n]


def ref_think_game(api: RefSearchApi, go_api, seed, board, komi, scoring, opts, max_moves=1000):
    """One fixed-seed game of the REFERENCE's ThinkBestMove against itself; returns the moves (-1 = resign)."""
    from sayuri_amd.engine import Game
    api.set_options(options(opts))
    net = api.lib.ref_net_new(b"")
    game = Game(board, komi, scoring, api_=go_api)
    search = api.lib.ref_search_new(game._h, net)
    api.lib.ref_seed(seed, seed + 77)
    moves = []
    while not game.info()[10] and len(moves) < max_moves:
        mv = api.lib.ref_search_think(search, game._h)
        moves.append(mv)
        assert game.play(mv)
    api.lib.ref_search_free(search)
    api.lib.ref_net_free(net)
    return moves


def records_close(a: bytes, b: bytes, rel=3e-5, abs_=2e-6, racy_records=()):
    """Compare two training-record texts token by token: integers and bit strings exactly, floats within tolerance
    (the reference binary is built with -ffast-math; sums of a few hundred floats round differently).
    racy_records: samples (counting only written ones) whose search met a single-candidate root.  The reference
    stops such a search from a polling thread after a timing-dependent handful of playouts, so that sample's
    policy / stddev / kld lines -- and, through the look-ahead averages, every sample's averaged value and score
    targets -- are not reproducible run to run on the reference itself; they are left out of the comparison."""
    la, lb = a.split(b"\n"), b.split(b"\n")
    if len(la) != len(lb):
        return f"{len(la)} vs {len(lb)} lines"
    for i, (x, y) in enumerate(zip(la, lb)):
        if x == y:
            continue
        line = i % 53
        rec = i // 53
        if racy_records and (line in (48, 50) or (rec in racy_records and line in (44, 51, 52)) or (rec + 1 in racy_records and line == 45)):
            continue
        if 6 <= line <= 43 or line == 46:  # hex planes, side to move, ownership string
            return f"record {i // 53} line {line + 1}: binary field differs"
        tx, ty = x.split(), y.split()
        if len(tx) != len(ty):
            return f"record {i // 53} line {line + 1}: {len(tx)} vs {len(ty)} values"
        fx, fy = np.array(tx, float), np.array(ty, float)
        # scale of the line, not of the single value: the averaged targets are sums of terms of alternating sign
        # ... and the windowed / discounted score targets cancel terms of board-size magnitude
        floor = {48: 2e-5, 50: 1e-3}.get(line, abs_)
        bad = np.abs(fx - fy) > floor + rel * max(np.abs(fx).max(), np.abs(fy).max())
        if bad.any():
            j = int(np.flatnonzero(bad)[0])
            return f"record {i // 53} line {line + 1} value {j}: {tx[j]} vs {ty[j]}"
    return None
