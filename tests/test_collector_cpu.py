"""The host side of the hot path -- staging, leaf-batch collector, ring pump, hand-out of results
(sayuri_amd/csrc/host/hip_forward_pipe.cc; reference src/neural/batch_forward_pipe.cc:7-193) -- on a box without a
GPU: tests/fake_hip/fake_hip.c stands in for the device side of include/sayuri_hip.h (loaded RTLD_GLOBAL ahead of the
real library, so the host library's calls resolve to it).  Its "network" is a cheap fixed function of each sample's own
planes, so every reply can be checked against the request that asked for it: what is tested is that hundreds of
concurrent blocking callers, ragged board sizes, partial batches, ring rotation and the wake tree never mix requests up.

Runs in a subprocess: symbol interposition needs a process that has not loaded the real device library yet."""
import os
import subprocess
import sys
import textwrap

import pytest

from _golden import Golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_SRC = os.path.join(ROOT, "tests", "fake_hip", "fake_hip.c")

DRIVER = textwrap.dedent(r"""
    import ctypes, sys, threading
    import numpy as np
    ctypes.CDLL(sys.argv[1], mode=ctypes.RTLD_GLOBAL)      # the fake device side wins symbol resolution
    from sayuri_amd.pipe import HipForwardPipe

    B, C = 19, 43
    def expected(planes, bs, off):
        grid = np.zeros((C, B, B), np.float32)
        grid[:, :bs, :bs] = planes.reshape(C, bs, bs)
        x = grid.reshape(C, B * B).astype(np.float64)
        w = 1 + (np.arange(C * B * B) % 7)
        s = float((x.ravel() * w).sum())
        prob = (x[off] + 0.5 * x[5] + off).reshape(B, B)[:bs, :bs].ravel()
        own = (x[7] - x[8]).reshape(B, B)[:bs, :bs].ravel()
        misc = np.float32(s * 0.002) - np.arange(15, dtype=np.float32) + np.float32(bs)
        tail = [np.float32(s * 0.001) + off, misc[0], misc[1], misc[2], misc[3], misc[8], misc[13], misc[14], off]
        return np.concatenate([prob, own, np.asarray(tail, np.float64)])

    def check(outs, cases, label):
        for (p, bs, off), got in zip(cases, outs):
            exp = expected(p, bs, off)
            assert got.shape == exp.shape, (label, got.shape, exp.shape)
            err = np.abs(got - exp).max()
            assert err <= 1e-3 * max(1.0, np.abs(exp).max()), (label, bs, off, err)

    rng = np.random.default_rng(int(sys.argv[3]))
    def make(n):
        cases = []
        for _ in range(n):
            bs = int(rng.choice([19, 19, 13, 9, 7]))
            cases.append((rng.integers(0, 4, size=(C, bs * bs)).astype(np.float32), bs, int(rng.integers(0, 5))))
        return cases

    pipe = HipForwardPipe(sys.argv[2], board_size=19, batch_size=16, fp16=True)
    assert pipe.GetNumWorkers() == 1
    total = 0
    for rnd in range(int(sys.argv[4])):
        n = int(rng.choice([1, 2, 15, 16, 17, 33, 100, 250]))
        cases = make(n)
        planes, bsz, offs = [c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases]
        check(pipe.Forward(planes, bsz, offsets=offs), cases, f"queue round {rnd} n={n}")   # n blocked callers
        if n <= 16:
            check(pipe.BatchForward(planes, bsz, offsets=offs), cases, f"batch round {rnd}")
        total += n
    # packed planes (bit planes + scalars, csrc/host/packed_planes.h): all-packed batches go to the device packed, mixed
    # batches are expanded by the pump; the replies must be those of the fp32 route
    def make_packable(n):
        cases = []
        for _ in range(n):
            bs = int(rng.choice([19, 19, 13, 9, 7]))
            p = np.zeros((C, bs * bs), np.float32)
            p[:C - 6] = rng.integers(0, 2, size=(C - 6, bs * bs))
            p[C - 6:] = rng.normal(size=(6, 1)).astype(np.float32)
            cases.append((p, bs, int(rng.integers(0, 5))))
        return cases
    for n in (1, 16, 37, 120):
        cases = make_packable(n)
        planes, bsz, offs = [c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases]
        check(pipe.ForwardPacked(planes, bsz, offsets=offs), cases, f"packed n={n}")
        check(pipe.ForwardPacked(planes, bsz, offsets=offs, mixed=True), cases, f"mixed n={n}")
        check(pipe.Forward(planes, bsz, offsets=offs), cases, f"fp32 of packable n={n}")
        total += 3 * n
    # two Python threads driving the queue at once (each call spawns its own callers), plus the async Submit path
    errs = []
    def worker(seed):
        try:
            r = np.random.default_rng(seed)
            for _ in range(6):
                cs = [(r.integers(0, 4, size=(C, 361)).astype(np.float32), 19, int(r.integers(0, 5))) for _ in range(40)]
                check(pipe.Forward([c[0] for c in cs], [19] * 40, offsets=[c[2] for c in cs]), cs, f"thread {seed}")
        except Exception as e:       # noqa: BLE001
            errs.append(repr(e))
    ths = [threading.Thread(target=worker, args=(s,)) for s in (1, 2, 3)]
    [t.start() for t in ths]; [t.join() for t in ths]
    assert not errs, errs
    eps, tot = pipe.netbench(64, 0.5)
    assert tot > 0
    pt = pipe.pump_times()
    assert pt["evals"] >= total + 3 * 6 * 40 + tot, pt
    assert pt["batches"] > 0 and pt["evals"] / pt["batches"] <= 16
    # smaller NN board: Construct() while idle, then the same checks on 13x13
    pipe.Construct(13, 8)
    B = 13
    cs = [(rng.integers(0, 4, size=(C, bs * bs)).astype(np.float32), bs, 0) for bs in (13, 9, 13, 5, 13, 13, 9, 13, 13)]
    check(pipe.Forward([c[0] for c in cs], [c[1] for c in cs], offsets=[0] * len(cs)), cs, "13x13")
    pipe.Destroy()
    print("collector ok", total, pt["batches"], pt["partial_batches"])
""")


@pytest.fixture(scope="module")
def fake_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fake_hip") / "libfake_hip.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-Wall", FAKE_SRC, "-o", out, "-lpthread"])
    return out


@pytest.mark.parametrize("delay_us,seed", [(0, 1), (200, 2), (3000, 3)])
def test_collector_keeps_requests_apart(fake_lib, tmp_weights_dir, delay_us, seed):
    """delay 0: batches finish before the next one is full (mostly partial batches); 200 us: steady state;
    3 ms: the ring fills up and callers park on the epoch futex."""
    weights = Golden("tiny_res", tmp_weights_dir).weights_path
    env = dict(os.environ, FAKE_HIP_DELAY_US=str(delay_us), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    rounds = "10" if delay_us < 3000 else "5"
    r = subprocess.run([sys.executable, "-c", DRIVER, fake_lib, weights, str(seed), rounds], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "collector ok" in r.stdout


SELFPLAY_DRIVER = textwrap.dedent(r"""
    import ctypes, glob, gzip, sys
    ctypes.CDLL(sys.argv[1], mode=ctypes.RTLD_GLOBAL)
    from sayuri_amd.pipe import HipForwardPipe
    from sayuri_amd import search as S
    pipe = HipForwardPipe(sys.argv[2], board_size=9, batch_size=16, fp16=True, waittime_ms=2)
    opts = dict(playouts=24, parallel_games=40, num_games=40, seed=5, dirichlet_noise=1, random_moves_factor=0.1,
                selfplay_query=["bkp:9:7:0.7", "bkp:7:9:0.3"], target_directory=sys.argv[3], game_threads=int(sys.argv[4]))
    st = S.selfplay(pipe, opts, move_cap=24, name_suffix="-cpu")
    pt = pipe.pump_times()
    assert st["games_done"] == 40 and st["chunks_saved"] == 40, st
    assert st["nn_queries"] == pt["evals"] > 2000, (st, pt)
    assert pt["evals"] / pt["batches"] > 4, "games are not being batched together"
    chunks = glob.glob(sys.argv[3] + "/tdata/*-cpu/*.gz")
    assert len(chunks) == 40
    lines = gzip.open(chunks[0]).read().decode().split("\n")
    assert (len(lines) - 1) % 53 == 0
    pipe.Destroy()
    print("selfplay ok", st["moves"], pt["batches"])
""")


@pytest.mark.parametrize("game_threads", [-1, 3], ids=["thread-per-game", "fibers-on-3-threads"])
def test_selfplay_through_the_collector_without_a_gpu(fake_lib, tmp_weights_dir, tmp_path, game_threads):
    """The whole host-side path of a self-play run -- game threads, search, encoder, NN cache, collector, training-data
    writer -- on the fake device: 40 games of 9x9 / 7x7 finish, every NN query went through the pump, the chunks have the
    53-line record format.  Once with one OS thread per game (the reference's scheme), once with the 40 games as fibers
    on 3 threads (csrc/host/fiber.h: a game yields inside Forward() instead of parking its thread)."""
    weights = Golden("tiny_res", tmp_weights_dir).weights_path
    env = dict(os.environ, FAKE_HIP_DELAY_US="300", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", SELFPLAY_DRIVER, fake_lib, weights, str(tmp_path), str(game_threads)], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "selfplay ok" in r.stdout


TWO_PUMP_DRIVER = textwrap.dedent(r"""
    import ctypes, sys, threading
    import numpy as np
    ctypes.CDLL(sys.argv[1], mode=ctypes.RTLD_GLOBAL)
    from sayuri_amd.pipe import HipForwardPipe
    B, C = 19, 43
    def expected(planes, bs, off):
        grid = np.zeros((C, B, B), np.float32)
        grid[:, :bs, :bs] = planes.reshape(C, bs, bs)
        x = grid.reshape(C, B * B).astype(np.float64)
        prob = (x[off] + 0.5 * x[5] + off).reshape(B, B)[:bs, :bs].ravel()
        return prob
    pipe = HipForwardPipe(sys.argv[2], board_size=19, batch_size=8, fp16=True, device=-1)   # -1: every visible GPU
    assert pipe.GetNumWorkers() == 2, pipe.GetNumWorkers()
    rng = np.random.default_rng(11)
    errs = []
    def worker(seed):
        try:
            r = np.random.default_rng(seed)
            for _ in range(8):
                n = int(r.choice([1, 7, 8, 9, 30]))
                cs = [(r.integers(0, 4, size=(C, bs * bs)).astype(np.float32), bs, int(r.integers(0, 5))) for bs in r.choice([19, 13, 9], size=n)]
                outs = pipe.Forward([c[0] for c in cs], [int(c[1]) for c in cs], offsets=[c[2] for c in cs])
                for (p, bs, off), got in zip(cs, outs):
                    exp = expected(p, int(bs), off)
                    assert np.abs(got[:bs * bs] - exp).max() <= 1e-3 * max(1.0, np.abs(exp).max())
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))
    ths = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    [t.start() for t in ths]; [t.join() for t in ths]
    assert not errs, errs
    pt = pipe.pump_times()
    assert pt["batches"] > 0
    # BatchForward addresses one GPU's graph directly (reference BatchForward(gpu, inputs))
    cs = [(rng.integers(0, 4, size=(C, 361)).astype(np.float32), 19, 0) for _ in range(5)]
    for gpu in (0, 1):
        outs = pipe.BatchForward([c[0] for c in cs], [19] * 5, offsets=[0] * 5, gpu=gpu)
        for (p, bs, off), got in zip(cs, outs):
            assert np.abs(got[:361] - expected(p, 19, 0)).max() <= 1e-3 * 10
    pipe.Destroy()
    print("two pumps ok", pt["batches"], pt["evals"])
""")


def test_in_process_two_gpu_pumps(fake_lib, tmp_weights_dir):
    """The drop-in's in-process multi-GPU form (reference: one NNGraph per --gpu inside one process, GetNumWorkers() =
    number of GPUs): with FAKE_HIP_DEVICES=2 the pipe builds two graphs, each with its own pump thread and staging ring;
    blocking callers are spread over both and every reply still belongs to its request."""
    weights = Golden("tiny_res", tmp_weights_dir).weights_path
    env = dict(os.environ, FAKE_HIP_DELAY_US="300", FAKE_HIP_DEVICES="2",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", TWO_PUMP_DRIVER, fake_lib, weights], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "two pumps ok" in r.stdout
