"""Shared by the golden generator (tests/golden/make_golden_games.py) and tests/test_engine_cpu.py:
the seeded game configurations and the move-choice policy, so that generator and test walk the same games."""
from __future__ import annotations

import hashlib

import numpy as np

GAME_CONFIGS = [
    dict(seed=11, board=19, komi=7.5, scoring=0, handicap=0, version=4, max_moves=420),
    dict(seed=12, board=19, komi=6.5, scoring=1, handicap=0, version=5, max_moves=380),
    dict(seed=13, board=19, komi=0.5, scoring=0, handicap=4, version=3, max_moves=360),
    dict(seed=14, board=13, komi=7.0, scoring=0, handicap=0, version=4, max_moves=260),
    dict(seed=15, board=9, komi=7.0, scoring=0, handicap=0, version=4, max_moves=160),
    dict(seed=16, board=9, komi=-3.5, scoring=1, handicap=2, version=2, max_moves=160),
    dict(seed=17, board=7, komi=9.0, scoring=0, handicap=0, version=1, max_moves=120),
    dict(seed=18, board=5, komi=24.0, scoring=0, handicap=0, version=4, max_moves=80),
    dict(seed=19, board=2, komi=0.5, scoring=0, handicap=0, version=4, max_moves=30),
    dict(seed=20, board=19, komi=7.5, scoring=0, handicap=9, version=4, max_moves=300),
]

RNG_SEEDS = (0, 1, 0xabcdabcd12345678, 2 ** 64 - 3)


def digest(a: np.ndarray) -> bytes:
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).digest()


def choose_move(rng, maps: np.ndarray, n: int, step: int):
    """(op, move): op 0 = play `move`, 1 = undo, 2 = set the territory helper from the ownership map.
    Mostly sensible random play (own real eyes are not filled) so games are long and full of captures,
    kos, ladders and pass-alive groups; a few wild moves, passes and undos."""
    r = rng.random()
    if r < 0.01 and step > 2:
        return 1, 0
    if r < 0.02:
        return 2, 0
    legal = np.flatnonzero(maps[1][:n])
    good = legal[(maps[8][legal] & 16) == 0]
    r = rng.random()
    if len(good) == 0 or r < 0.02:
        return 0, n
    if r < 0.05:
        return 0, int(rng.choice(legal))
    return 0, int(rng.choice(good))
