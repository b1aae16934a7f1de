"""Go engine (board / game state / encoder) parity, CPU only.

* golden replay: tests/golden/go_games.npz holds games recorded from the REFERENCE engine
  (generator: tests/golden/make_golden_games.py).  The product engine must reproduce, move by move,
  every state word (Zobrist hashes incl. the 8 symmetry hashes, ko, prisoners, superko ...), every scalar
  (komi with penalty, wave, final score) and the SHA-1 of the analysis maps (legality, liberties, ladder
  codes, pass-alive safe area, ownership, seki, move tactics) and of the 43/38 input planes.  Bit-exact.
* live: where oracle/_ref is available (dev container) fresh random games are compared directly.
"""
import ctypes
import os

import numpy as np
import pytest

from go_replay import GAME_CONFIGS, RNG_SEEDS, choose_move, digest
from sayuri_amd import _lib
from sayuri_amd.engine import INFO_NAMES, MAP_NAMES, Game, GoApi, expand_packed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "go_games.npz")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libsayuri_ref.so")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


@pytest.mark.parametrize("gi", range(len(GAME_CONFIGS)))
def test_golden_replay(golden, gi):
    cfg = GAME_CONFIGS[gi]
    g = Game(cfg["board"], cfg["komi"], cfg["scoring"])
    if cfg["handicap"]:
        assert g.fixed_handicap(cfg["handicap"])
    moves, infos = golden[f"g{gi}_moves"], golden[f"g{gi}_info"]
    scalars, digests, symms = golden[f"g{gi}_scalars"], golden[f"g{gi}_digest"], golden[f"g{gi}_symm"]
    assert len(infos) >= len(moves)
    for step in range(len(infos)):
        info, sc, maps = g.info(), g.scalars(), g.maps()
        planes = g.planes(int(symms[step]), cfg["version"])
        for k, name in enumerate(INFO_NAMES):
            assert info[k] == infos[step][k], f"game {gi} step {step}: {name}"
        assert np.array_equal(sc, scalars[step]), f"game {gi} step {step}: scalars {sc} vs {scalars[step]}"
        assert digest(maps) == digests[step][:20].tobytes(), f"game {gi} step {step}: analysis maps differ"
        assert digest(planes) == digests[step][20:].tobytes(), f"game {gi} step {step}: input planes differ"
        # the compact encoder (bit planes + broadcast scalars) expands to the same 43 / 38 planes, bit for bit
        rec, binary = g.planes_packed(int(symms[step]), cfg["version"])
        assert binary == (34 if cfg["version"] in (1, 2) else 37)
        expanded = expand_packed(rec, binary, cfg["board"], planes.shape[0])
        assert np.array_equal(expanded.view(np.uint32), planes.view(np.uint32)), f"game {gi} step {step}: packed planes differ"
        if step < len(moves):
            op, move = int(moves[step][0]), int(moves[step][1])
            if op == 0:
                assert g.play(move)
            elif op == 1:
                assert g.undo()
            else:
                g.set_territory_helper_from_ownership()


def test_rng_known_answers(golden):
    lib = _lib.host()
    lib.sayuri_go_rng_stream.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, ctypes.c_double, ctypes.c_void_p]
    for si, seed in enumerate(RNG_SEEDS):
        buf = np.zeros(3 * 64, np.uint64)
        lib.sayuri_go_rng_stream(seed, 64, 362, 0.37, buf.ctypes.data)
        assert np.array_equal(buf, golden[f"rng{si}"])


def test_illegal_moves_and_edges():
    g = Game(9, 7.0)
    assert g.play(40)
    assert not g.play(40)            # occupied
    assert g.undo()
    assert g.info()[8] == 0
    assert not g.undo()              # nothing left to undo
    # suicide: a white stone into a corner whose two neighbours are black
    g = Game(5, 0.0)
    for mv in (1, 5):
        assert g.play(mv, 0)
    assert not g.play(0, 1)
    assert g.play(0, 0)              # filling one's own eye is legal
    # two passes end the game; resigning sets the winner
    g = Game(9, 7.0)
    assert g.play(81) and g.play(81)
    assert g.info()[10] == 1
    g = Game(9, 7.0)
    assert g.play(-1)
    assert g.info()[12] == 1 and g.info()[10] == 1   # black resigned: white wins
    # komi must be an integer or a half
    g = Game(9, 7.0)
    g.set_komi(6.25)
    assert g.scalars()[0] == 7.0


def test_ko_and_superko():
    # 5x5, index = y*5+x.  Black 1,5,11 / white 2,8,12 and a white stone at 6 in atari; black takes at 7 -> ko at 6
    g = Game(5, 0.0)
    for mv, c in ((1, 0), (5, 0), (11, 0), (2, 1), (8, 1), (12, 1), (6, 1)):
        assert g.play(mv, c), mv
    assert g.play(7, 0)
    info, maps = g.info(), g.maps()
    assert info[6] == 1 and info[4] == 6 and maps[0][6] == 2   # one prisoner, ko point, stone lifted
    assert maps[1][6] == 0 and not g.play(6, 1)                # white may not retake at once
    assert g.play(20, 1) and g.play(24, 0)
    assert g.info()[4] == np.uint64(2 ** 64 - 1)               # ko cleared
    assert g.play(6, 1)                                        # now white retakes: the earlier position repeats
    assert g.info()[9] == 0                                    # ... not yet (extra stones at 20/24 differ)
    assert g.play(25) and g.play(25)                           # passes


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref is only built in the dev container")
@pytest.mark.parametrize("seed,board,komi,scoring,handicap,version",
                         [(101, 19, 7.5, 0, 0, 4), (102, 19, 5.5, 1, 3, 4), (103, 13, 7.5, 0, 0, 2), (104, 9, 7.0, 0, 0, 4),
                          (105, 6, 4.0, 0, 0, 4), (106, 4, 2.0, 1, 0, 4), (107, 19, 7.5, 0, 0, 4), (108, 9, 6.0, 1, 0, 5),
                          (109, 19, 6.5, 1, 0, 4), (110, 19, 7.5, 0, 2, 3), (111, 13, 5.5, 1, 0, 4), (112, 19, 0.5, 0, 0, 5)])
def test_live_against_reference(seed, board, komi, scoring, handicap, version):
    ref = GoApi(ctypes.CDLL(REF_SO), "ref_game_")
    a, b = Game(board, komi, scoring, api_=ref), Game(board, komi, scoring)
    if handicap:
        assert a.fixed_handicap(handicap) == b.fixed_handicap(handicap)
    rng = np.random.default_rng(seed)
    for step in range(450):
        symm = int(rng.integers(8))
        ia, ib = a.info(), b.info()
        assert np.array_equal(ia, ib), (step, ia, ib)
        assert np.array_equal(a.scalars(), b.scalars()), step
        ma, mb = a.maps(), b.maps()
        for k, name in enumerate(MAP_NAMES):
            assert np.array_equal(ma[k], mb[k]), (step, name)
        assert np.array_equal(a.planes(symm, version), b.planes(symm, version)), step
        if ia[10]:
            break
        op, move = choose_move(rng, ma, a.n, step)
        if op == 0:
            assert a.play(move) and b.play(move)
        elif op == 1:
            assert a.undo() == b.undo()
        else:
            a.set_territory_helper_from_ownership()
            b.set_territory_helper_from_ownership()
