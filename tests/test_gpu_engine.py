"""Search and self-play on the MI355X forward pipe.

* fixed-seed search, fp32 engine: the golden games of tests/golden/search_games.npz were played by the REFERENCE
  search on the reference's CPU pipe; the product engine on HipForwardPipe (fp32, abs <= 1e-4 to that pipe) must
  choose the same moves and emit the same training records (floats within the fp32 tolerance of the network).
* fp16 engine: same positions, the root visit distribution must stay close and the best move equal.
* self-play loop on the GPU: many concurrent games feeding one batched queue; records and SGF come out well-formed.
"""
import glob
import gzip
import os
import zlib

import numpy as np
import pytest

from search_replay import NN_GAMES, options, records_close
from sayuri_amd import search as S
from sayuri_amd import weights as W
from sayuri_amd.engine import Game
from sayuri_amd.pipe import HipForwardPipe

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def w6b96(tmp_weights_dir):
    path = os.path.join(tmp_weights_dir, "engine_6b96.bin")
    if not os.path.exists(path):
        W.write_weights(path, W.spec_6b96(), seed=21)
    return path


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "search_games.npz"))


def play(net, seed, board, komi, scoring, opts, nmoves):
    game = Game(board, komi, scoring)
    search = S.Search(game, net, options(opts), seeds=(seed, seed + 77))
    moves = []
    while not game.info()[10] and len(moves) < nmoves:
        mv = search.selfplay_move()
        moves.append(mv)
        assert game.play(mv)
    search.update_territory_helper()
    racy = search.single_candidate_records()
    return moves, search.gather_training_text(), racy


@pytest.mark.parametrize("i", range(len(NN_GAMES)))
def test_fixed_seed_search_reproduces_reference_moves_fp32(golden, w6b96, i):
    seed, board, komi, scoring, opts, nmoves = NN_GAMES[i]
    pipe = HipForwardPipe(w6b96, board_size=board, batch_size=8, fp16=False, waittime_ms=0)
    net = S.Network(pipe=pipe, options=options(opts))
    moves, text, racy = play(net, seed, board, komi, scoring, opts, nmoves)
    assert moves == golden[f"nn{i}_moves"].tolist()
    # network outputs agree to 1e-4 abs (fp32); the records' policy / value targets inherit that
    assert records_close(zlib.decompress(golden[f"nn{i}_records"].tobytes()), text, rel=2e-3, abs_=2e-4, racy_records=racy) is None
    assert net.queries() > 0
    pipe.Destroy()


def test_fp16_search_close_to_fp32(w6b96):
    results = {}
    for fp16 in (False, True):
        pipe = HipForwardPipe(w6b96, board_size=9, batch_size=8, fp16=fp16, waittime_ms=0)
        net = S.Network(pipe=pipe, options=options({}))
        game = Game(9, 7.0, 0)
        for mv in (40, 30, 50, 22):
            assert game.play(mv)
        search = S.Search(game, net, options(dict(playouts=200)), seeds=(5, 6))
        results[fp16] = search.computation(200, S.TAG_UNREUSED)
        search.close()
        pipe.Destroy()
    a, b = results[False], results[True]
    assert a["visits"] == b["visits"] == 201
    assert a["best_move"] == b["best_move"]
    # visit distributions: total variation distance
    tv = 0.5 * np.abs(a["root_visits"] / 200.0 - b["root_visits"] / 200.0).sum()
    assert tv < 0.15, tv
    assert abs(a["root_eval"] - b["root_eval"]) < 0.02


def test_selfplay_on_the_gpu_queue(w6b96, tmp_path):
    pipe = HipForwardPipe(w6b96, board_size=9, batch_size=32, fp16=True, waittime_ms=2)
    opts = dict(playouts=48, parallel_games=64, num_games=64, seed=11, dirichlet_noise=1, first_pass_bonus=1, random_moves_factor=0.1,
                komi_stddev=2.5, selfplay_query=["bkp:9:7:0.8", "bkp:7:9:0.2"], early_symm_cache=1, target_directory=str(tmp_path))
    st = S.selfplay(pipe, opts, move_cap=40, name_suffix="-r0")
    pt = pipe.pump_times()
    assert st["games_done"] == 64 and st["chunks_saved"] == 64
    assert st["nn_queries"] == pt["evals"] > 10000
    assert pt["evals"] / pt["batches"] > 8, "games are not being batched together"
    assert st["cache_hits"] > 0
    chunks = glob.glob(str(tmp_path / "tdata" / "*-r0" / "*.gz"))
    assert len(chunks) == 64
    lines = gzip.open(chunks[0]).read().decode().split("\n")
    assert (len(lines) - 1) % 53 == 0
    assert open(glob.glob(str(tmp_path / "sgf" / "*.sgf"))[0]).read().count("(;GM[1]") == 64
    pipe.Destroy()


def test_netbench_agrees_with_the_pump_counters(w6b96):
    """The reference's `netbench` (src/game/gtp.cc:1468-1568: threads hammering Forward with the cache off, evals = pipe
    calls / wall): every evaluation it counts went through the pump, and the rate it reports is the counted total over the
    window."""
    pipe = HipForwardPipe(w6b96, board_size=19, batch_size=64, fp16=True, waittime_ms=2)
    before = pipe.pump_times()
    eps, total = pipe.netbench(threads=128, seconds=1.5)
    after = pipe.pump_times()
    pumped = after["evals"] - before["evals"]
    assert total > 1000 and eps > 0
    # threads still inside Forward() when the window closes finish their evaluation after the count was taken
    assert total <= pumped <= total + 128, (total, pumped)
    assert abs(eps - total / 1.5) <= 0.1 * eps
    batches = after["batches"] - before["batches"]
    assert pumped / batches > 32, "netbench callers are not being batched"
    pipe.Destroy()


def test_search_benchmark_mode_on_the_gpu(w6b96):
    """Reference --mode benchmark (src/benchmark/benchmark.cc:110-161) on the HIP pipe: one search at a time gives the
    latency figure, 64 concurrent searches the throughput figure; every playout beyond the root costs one evaluation
    at most (cache off)."""
    pipe = HipForwardPipe(w6b96, board_size=9, batch_size=32, fp16=True, waittime_ms=1)
    one = S.benchmark(pipe, dict(playouts=100, default_boardsize=9, seed=3), positions=4, concurrent=1)
    many = S.benchmark(pipe, dict(playouts=100, default_boardsize=9, seed=3), positions=64, concurrent=64)
    assert one["playouts_per_move"] == 100 and many["playouts_per_move"] == 100
    # cache off: the root, one evaluation per playout, and the few extra root-level queries of a search (ownership)
    assert 0 < one["nn_queries"] <= 4 * 104 and 0 < many["nn_queries"] <= 64 * 104
    assert many["playouts_per_second_total"] > 3 * one["playouts_per_second_total"], (one, many)
    pipe.Destroy()


def test_selfplay_configs2_workload_keeps_full_batches(tmp_weights_dir, tmp_path):
    """BASELINE.json configs[2] at its real size for half a minute: 20b256, 19x19, 512 concurrent games, 400 visits, batch
    256.  The queue must run full (mean batch > 200), records must be well formed, nothing may fail."""
    path = os.path.join(tmp_weights_dir, "engine_20b256.bin")
    if not os.path.exists(path):
        W.write_weights(path, W.spec_20b256(), seed=22)
    pipe = HipForwardPipe(path, board_size=19, batch_size=256, fp16=True, waittime_ms=2)
    opts = dict(playouts=400, parallel_games=512, num_games=1000000, seed=77, dirichlet_noise=1, dirichlet_epsilon=0.25, dirichlet_init=0.03,
                dirichlet_factor=361, first_pass_bonus=1, random_moves_factor=0.1, komi_stddev=2.5, resign_playouts=80, resign_threshold=0.05,
                early_symm_cache=1, cache_memory_mib=400, selfplay_query=["bkp:19:7:1"], stagger_moves=360, target_directory=str(tmp_path))
    st = S.selfplay(pipe, opts, seconds=30.0, name_suffix="-r0")
    pt = pipe.pump_times()
    assert st["nn_queries"] > 30 * 20000, st          # > 20 k evals/s through encoder + queue + PCIe
    assert pt["evals"] / pt["batches"] > 200, (pt["evals"], pt["batches"])
    assert st["moves"] > 1000 and st["playouts"] > 100000
    assert st["games_done"] > 0 and st["chunks_saved"] == st["games_done"], st   # staggered starts: games do finish in the window
    chunks = glob.glob(str(tmp_path / "tdata" / "*-r0" / "*.gz"))
    assert len(chunks) == st["games_done"]
    lines = gzip.open(chunks[0]).read().decode().split("\n")
    assert (len(lines) - 1) % 53 == 0 and lines[1] == "0"   # version line, then the mode line of a 19x19 record
    pipe.Destroy()


def test_bench_two_ranks_share_the_gpu():
    """bench.py's multi-rank path against the real runtime: `python bench.py --gpus 2` (the script starts its two ranks, one
    process per rank), on gloo, the two ranks sharing this box's one GPU (SAYURI_BENCH_SHARE_DEVICE).  What is checked is
    that the path runs -- barrier, max-over-ranks timing, the stats gather, the exchange rounds of the self-play window, ONE
    JSON line from rank 0 -- not the throughput of two processes on one device (tests/test_dropin_cpu.py runs the same on the
    fake device with N = 2; profiles/r04_fake8_bench.json with N = 8)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SAYURI_BENCH_SHARE_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    # round 6: the PLAIN command -- no launcher around it; bench.py starts its two ranks itself (launch_ranks), and configs[4]
    # runs on both of them
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--master-port", "29547", "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--dist-backend", "gloo", "--selfplay-seconds", "10", "--selfplay-games", "64", "--selfplay-visits", "16"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
    sp = d["selfplay"]
    assert sp["exchange_rounds"] > 0 and sp["nn_evals_per_sec"] > 0 and not sp["halt_seen"]
    assert "cpu_baseline" not in d
    assert d["config5"]["n_gpus"] == 2 and len(d["config5"]["per_rank_evals_per_sec"]) == 2 and d["tower"] == "persistent"


def test_bench_exchange_over_rccl_beside_the_persistent_launch():
    """The path's one collective on the backend the 8-GPU run uses: bench.py with torch.distributed initialised on "nccl"
    (= RCCL) for a world of one, so that the start / stop barriers, the stats all-gather and the periodic exchange of the
    self-play window go through RCCL kernels and small copies on a torch stream WHILE the two tickets' persistent tower
    launches hold every CU of the device.  The exchange must land every round, without holding the
    window up, and the writer must have put every finished game on disk."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29551", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--force-dist", "--dist-backend", "nccl",
           "--selfplay-seconds", "30", "--no-cpu-baseline", "--no-config5", "--no-pump"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    sp = d["selfplay"]
    ex = sp["exchange"]
    print("exchange over RCCL beside the tower launches:", ex, "self-play evals/s", sp["nn_evals_per_sec"], "microbench", d["value"])
    assert ex["backend"] == "nccl" and ex["world"] == 1
    assert sp["exchange_rounds"] >= 10 and ex["rounds"] == sp["exchange_rounds"]      # a round every 2 s of a 30 s window
    assert ex["late_rounds"] == 0 and ex["skipped_ticks"] == 0 and ex["max_ms"] < 1000.0, ex
    assert sp["mean_batch"] > 200 and sp["nn_evals_per_sec"] > 0.85 * d["value"], sp
    assert sp["games_done"] > 0 and sp["chunks_saved"] == sp["games_done"] and sp["bytes_written"] > 0, sp


def test_configs0_game_at_its_stated_size_fp32(golden, w6b96):
    """BASELINE.json configs[0] at the size it states -- 9x9, 6b x 96 network, 100 visits per move, one playout at a time --
    played by the reference search on the reference's CPU pipe (tests/golden/make_golden_search.py, NN_GAMES[3]); the product
    search on the fp32 HIP pipe must choose the same ten moves and write the same records."""
    seed, board, komi, scoring, opts, nmoves = NN_GAMES[3]
    assert opts["playouts"] == 100 and board == 9
    pipe = HipForwardPipe(w6b96, board_size=board, batch_size=8, fp16=False, waittime_ms=0)
    net = S.Network(pipe=pipe, options=options(opts))
    moves, text, racy = play(net, seed, board, komi, scoring, opts, nmoves)
    assert moves == golden["nn3_moves"].tolist()
    assert records_close(zlib.decompress(golden["nn3_records"].tobytes()), text, rel=2e-3, abs_=2e-4, racy_records=racy) is None
    pipe.Destroy()
