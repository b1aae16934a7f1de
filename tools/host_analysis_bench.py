#!/usr/bin/env python3
"""Host cost of the per-position board analyses, without a device: replay seeded 19x19 games (the move policy of the golden
games, tests/go_replay.py) and time, at every position, the ladder map, the score / pass-alive analysis and the packed
encoding (sayuri_go_encode_seconds, go_capi.cc).  Prints mean microseconds per position by game phase.

    python tools/host_analysis_bench.py [--games 4] [--moves 330]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import go_replay  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sayuri_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4)
    ap.add_argument("--moves", type=int, default=330)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--kinds", default="3,2,1", help="3 = score + pass-alive, 2 = ladder map, 1 = packed encoding")
    ap.add_argument("--from-move", type=int, default=0, help="time only positions from this move on")
    ap.add_argument("--symmetry", type=int, default=0, help="symmetry of the encodings (self-play draws one of 8 per evaluation)")
    ap.add_argument("--repeat", type=int, default=5, help="each position is timed this many times, the fastest counts (shared hosts)")
    a = ap.parse_args()
    lib = _lib.host()
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.sayuri_go_new.restype, lib.sayuri_go_new.argtypes = vp, [ci, ctypes.c_float, ci]
    lib.sayuri_go_play.restype, lib.sayuri_go_play.argtypes = ci, [vp, ci, ci]
    lib.sayuri_go_maps.restype, lib.sayuri_go_maps.argtypes = None, [vp, vp]
    lib.sayuri_go_free.restype, lib.sayuri_go_free.argtypes = None, [vp]
    lib.sayuri_go_encode_seconds.restype, lib.sayuri_go_encode_seconds.argtypes = ctypes.c_double, [vp, ci, ci, ci, ci]
    n = 361
    phases = [(0, 60), (60, 150), (150, 240), (240, 10 ** 6)]
    acc = {(k, ph): [] for k in (1, 2, 3) for ph in range(len(phases))}
    for gi in range(a.games):
        rng = np.random.default_rng(1000 + gi)
        h = lib.sayuri_go_new(19, 7.5, 0)
        maps = np.zeros((9, n + 1), np.uint8)
        for step in range(a.moves):
            lib.sayuri_go_maps(h, maps.ctypes.data)
            op, mv = go_replay.choose_move(rng, maps, n, step)
            if op != 0 or not lib.sayuri_go_play(h, mv, -1):
                continue
            ph = next(i for i, (lo, hi) in enumerate(phases) if lo <= step < hi)
            if step < a.from_move:
                continue
            for kind in [int(k) for k in a.kinds.split(",")]:
                acc[(kind, ph)].append(min(lib.sayuri_go_encode_seconds(h, a.iters, kind, a.symmetry, 4) for _ in range(a.repeat)) / a.iters * 1e6)
        lib.sayuri_go_free(h)
    names = {3: "score + pass-alive", 2: "ladder map", 1: "packed encoding (areas cached)"}
    for kind in [int(k) for k in a.kinds.split(",")]:
        row = "  ".join("moves %3d-%-4s %6.2f us" % (phases[ph][0], phases[ph][1] if phases[ph][1] < 10 ** 6 else "", float(np.mean(acc[(kind, ph)])))
                        for ph in range(len(phases)) if acc[(kind, ph)])
        print("%-32s %s   all %6.2f us" % (names[kind], row, float(np.mean(sum((acc[(kind, ph)] for ph in range(len(phases))), [])))))


if __name__ == "__main__":
    main()
