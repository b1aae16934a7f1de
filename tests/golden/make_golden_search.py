#!/usr/bin/env python3
"""Generate tests/golden/search_games.npz -- fixed-seed self-play games of the REFERENCE search.

Dev container only: drives the reference's Network / Search / TrainingData through
oracle/_ref/libsayuri_ref.so (taps in oracle/ref_search_driver.cc) with its random generators pinned
(ref_seed).  Recorded per game: the moves and the 53-line training records the reference wrote.
tests/test_search_cpu.py replays the same seeds through the product engine.

    python tests/golden/make_golden_search.py [--append]

--append keeps the arrays search_games.npz already holds and plays only the games it lacks (a new entry in
tests/search_replay.py then leaves the committed vectors of the others byte for byte as they were).
"""
from __future__ import annotations

import ctypes
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from sayuri_amd import weights as W  # noqa: E402
from sayuri_amd.engine import GoApi  # noqa: E402
from search_replay import DUMMY_GAMES, NN_GAMES, REF_SO, THINK_GAMES, RefSearchApi, ref_selfplay_game, ref_think_game  # noqa: E402


def main():
    api = RefSearchApi()
    go_api = GoApi(api.lib, "ref_game_")
    out = {}
    path = os.path.join(HERE, "search_games.npz")
    if "--append" in sys.argv and os.path.exists(path):
        out = dict(np.load(path))
    for i, (seed, board, komi, scoring, opts) in enumerate(DUMMY_GAMES):
        if f"dummy{i}_moves" in out:
            continue
        moves, text = ref_selfplay_game(api, go_api, seed, board, komi, scoring, opts)
        out[f"dummy{i}_moves"] = np.array(moves, np.int16)
        out[f"dummy{i}_records"] = np.frombuffer(zlib.compress(text, 9), np.uint8)
        print(f"dummy game {i}: board {board}, {len(moves)} moves, {len(text)} bytes of records", file=sys.stderr)
    for i, (seed, board, komi, scoring, opts) in enumerate(THINK_GAMES):
        if f"think{i}_moves" in out:
            continue
        moves = ref_think_game(api, go_api, seed, board, komi, scoring, opts)
        out[f"think{i}_moves"] = np.array(moves, np.int16)
        print(f"think game {i}: board {board}, {len(moves)} moves, last {moves[-1]}", file=sys.stderr)
    wpath = "/tmp/sayuri_golden_6b96_seed21.bin"
    W.write_weights(wpath, W.spec_6b96(), seed=21)
    assert api.lib.ref_init(wpath.encode(), 1) == 0
    for i, (seed, board, komi, scoring, opts, nmoves) in enumerate(NN_GAMES):
        if f"nn{i}_moves" in out:
            continue
        moves, text = ref_selfplay_game(api, go_api, seed, board, komi, scoring, opts, weights=wpath.encode(), max_moves=nmoves)
        out[f"nn{i}_moves"] = np.array(moves, np.int16)
        out[f"nn{i}_records"] = np.frombuffer(zlib.compress(text, 9), np.uint8)
        print(f"nn game {i}: board {board}, {len(moves)} moves, {len(text)} bytes of records", file=sys.stderr)
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
