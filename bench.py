#!/usr/bin/env python3
"""bench.py -- NN evals/sec of the Sayuri forward pipe on MI355X.

Workload = BASELINE.json configs[1]: 19x19 board, 20-block x 256-filter network (SE on every
3rd block, 32-channel heads, Mish), batch = 256, inference only, synthetic planes, random-init
weights (seeded).  A "step" is one pass of the whole network over one batch of 256 positions
whose input planes are already resident in HBM; value = whole-job evals/s over all ranks.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; positions (games) shard embarrassingly, there is no data-path
collective -- torch.distributed (RCCL) is used only for the start/stop barrier, the
max-over-ranks timing and the stats gather.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GFLOP_PER_EVAL_20B256 = None  # computed from the layer list below
PEAK_TFLOPS = {"f16": 2500.0, "f32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md


def algorithmic_flops_per_eval(spec, board: int = 19) -> float:
    """2*MAC of the direct convolutions + FCs of one evaluation."""
    s = board * board
    c = spec.channels
    mac = 43 * c * 9 * s
    for b in spec.blocks:
        if b.kind == "ResidualBlock":
            mac += 2 * c * c * 9 * s
        elif b.kind == "BottleneckBlock":
            i = b.bottleneck_channels or c // 2
            mac += 2 * c * i * s + 2 * i * i * 9 * s
        elif b.kind == "NestedBottleneckBlock":
            i = b.bottleneck_channels or c // 2
            mac += 2 * c * i * s + 4 * i * i * 9 * s
        elif b.kind == "MixerBlock":
            f = b.ffn_channels or int(1.5 * c)
            mac += c * b.kernel_size ** 2 * s + 2 * c * f * s
        if b.se:
            se = c // spec.se_ratio
            mac += 3 * c * se + se * 2 * c
    pc, vc = spec.policy_channels, spec.value_channels
    mac += c * pc * s + 3 * pc * pc + pc * 5 * s + pc * 5
    mac += c * vc * s + 9 * vc * vc + vc * s + 3 * vc * 15
    return 2.0 * mac


def hbm_traffic(fp16: bool, dominant: str) -> dict:
    """HBM-side (fabric) bytes per launch of the dominant kernel, from the rocprofv3 PMC passes kept under profiles/
    (counters cannot be read from inside this process; MI355X_MICROARCH.md HBM section: separate --pmc passes, requests x
    calibrated bytes per request).  Round 4: the passes run on the persistent tower launch itself (conv_tower_kernel<4>;
    rounds 2-3 had to sum per-layer launches).  `traffic_source` names the file, its age and the commit it was measured on."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic.json")))
    path = found[-1] if found else os.path.join(ROOT, "profiles", "r04_hbm_traffic.json")  # the latest round's passes
    layer_algo = 2 * 256 * 361 * 256 * 2 + 256 * 256 * 9 * 2  # in + out + weights of one 256->256 layer, fp16, batch 256
    if not fp16 or dominant != "tower_run" or not os.path.exists(path):
        return {"traffic": None, "algorithmic_bytes": layer_algo}
    t = json.load(open(path))
    rd, wr, algo = t["tower_run"]["read_bytes"], t["tower_run"]["write_bytes"], t["tower_run"]["algorithmic_bytes"]
    age_h = (time.time() - os.path.getmtime(path)) / 3600.0
    return {"traffic": rd + wr, "traffic_read": rd, "traffic_write": wr, "algorithmic_bytes": algo,
            "traffic_over_algorithmic": round((rd + wr) / algo, 3), "traffic_is_per_layer_sum": False,
            "traffic_source": f"{os.path.relpath(path, ROOT)} ({t['source']}; measured on commit {t.get('commit', '?')}, file {age_h:.1f} h old)"}


def long_run_games_per_hour() -> dict:
    """games/hour by the reference's definition with nothing helping it (finished games / wall, every game from the EMPTY
    board, more than one generation of the 512 games, the writer with the reference's pool; src/selfplay/pipe.cc:272-280) takes
    a 27-minute run (tools/selfplay_bench.py --seconds 1620), not a window of this script: read from the latest such profile
    kept under profiles/, the way `roofline.traffic` is read, with the file's age and the tree it was measured on."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_selfplay_27min_512games*.json")),
                   key=lambda f: (os.path.basename(f)[:3], os.path.getmtime(f)))
    if not found:
        return {"games_per_hour_empty_board_27min": None}
    path = found[-1]
    d = json.load(open(path))
    age_h = (time.time() - os.path.getmtime(path)) / 3600.0
    return {"games_per_hour_empty_board_27min": d.get("games_per_hour"),
            "games_per_hour_empty_board_27min_source": f"{os.path.relpath(path, ROOT)}: {d.get('games_done')} games finished in "
            f"{d.get('elapsed', 0):.0f} s from the empty board at {d.get('nn_evals_per_sec')} NN evals/s, mean batch {d.get('mean_batch')}, "
            f"{d.get('host_cpu_cores_busy')} host cores, {d.get('chunks_saved')} chunks written (measured on commit "
            f"{d.get('commit', '?')}, file {age_h:.1f} h old; NOT measured in this run)"}


def toolchain_line() -> str:
    """First line of `hipcc --version` (the persistent tower kernel's seam is validated against one compiler)."""
    try:
        from sayuri_amd import _build
        return _build._toolchain().splitlines()[0].strip()
    except Exception as e:  # noqa: BLE001
        return "unknown (%r)" % (e,)


def mark_dominant(lib, ctx) -> str:
    """Mark the dominant kernel class for per-launch event timing inside the timed region: the persistent tower launch
    (one launch per run of board convolutions, conv_tower.h) when the engine uses it, else the per-layer tower convolutions
    (one event pair per run of five launches: an event is a barrier packet, one pair per launch costs ~5 % of the step)."""
    ms = ctypes.c_float(0)
    lib.sayuri_hip_mark_kernel(ctx, b"tower_run")
    lib.sayuri_hip_time_runs(ctx, 1, ctypes.byref(ms))
    from sayuri_amd import _lib
    stat = _lib.KernelStat()
    lib.sayuri_hip_timed_stat(ctx, ctypes.byref(stat))
    if stat.launches > 0:
        return "tower_run"
    lib.sayuri_hip_mark_kernel(ctx, b"conv3x3_tower/5")
    return "conv3x3_tower"


def config5_segment(lib, local_rank: int, rank: int, steps: int, warmup: int, device: int = None):
    """BASELINE.json configs[4] on this GPU: 40-block x 384-filter net, fp16, a batch of 256 samples whose board size is
    drawn uniformly from 9 / 13 / 19, planes resident in HBM.  Returns evals/s, the tower convolution's
    launch time and the MFMA fraction on the batch's REAL pixels (a 9x9 sample costs 81/361 of a 19x19 one)."""
    import ctypes
    from sayuri_amd import _lib
    from sayuri_amd import weights as W
    from sayuri_amd.pipe import HipForwardPipe
    spec = W.spec_40b384()
    wdir = f"/tmp/sayuri_bench_weights_{os.getuid()}"
    wpath = os.path.join(wdir + "_c5", "net_40b384_seed23.bin")
    if local_rank == 0 and not os.path.exists(wpath):
        os.makedirs(os.path.dirname(wpath), exist_ok=True)
        W.write_weights(wpath + ".tmp", spec, seed=23)
        os.replace(wpath + ".tmp", wpath)
    while not os.path.exists(wpath):
        time.sleep(0.2)
    n = 256
    rng = np.random.default_rng(5000 + rank)
    bsz = rng.choice([9, 13, 19], size=n).astype(np.int32)  # arrival order; the engine groups the samples by size on the device
    planes = W.synthetic_planes(n, [int(b) for b in bsz], seed=5100 + rank)
    grid = np.zeros((n, 43, 19, 19), np.float32)
    for i, (p, b) in enumerate(zip(planes, bsz)):
        grid[i, :, :b, :b] = p.reshape(43, b, b)
    grid = np.ascontiguousarray(grid.reshape(n, 43, 361))
    pipe = HipForwardPipe(wpath, board_size=19, batch_size=n, fp16=True, device=local_rank if device is None else device)
    ctx = pipe.ctx(0)
    if lib.sayuri_hip_upload(ctx, n, grid.ctypes.data_as(_lib.c_float_p), bsz.ctypes.data_as(_lib.c_int_p)):
        raise RuntimeError(lib.sayuri_hip_last_error().decode())
    ms = ctypes.c_float(0)
    # the throughput: forwards as the engine runs them -- for this network a batch is cut into chains of per-layer launches over
    # groups of board tiles, on streams of their own (Engine::forward: a layer of the whole batch is 450
    # workgroups = two rounds of the 256 CUs, the second 76 % full; chains let a group's next layer start on the CUs another
    # group's round leaves free; bit-identical to the one-chain forward, tests/test_gpu_net.py::test_chained_forward)
    lib.sayuri_hip_mark_kernel(ctx, b"")
    lib.sayuri_hip_time_runs(ctx, warmup, ctypes.byref(ms))
    lib.sayuri_hip_sync(ctx)
    t0 = time.perf_counter()
    if lib.sayuri_hip_time_runs(ctx, steps, ctypes.byref(ms)):
        raise RuntimeError(lib.sayuri_hip_last_error().decode())
    lib.sayuri_hip_sync(ctx)
    el = time.perf_counter() - t0
    chains = int(lib.sayuri_hip_last_chains(ctx))
    # the dominant kernel's own duration: a second pass with event pairs around its launches, as ONE chain (a launch that shares
    # the chip with another chain's has no duration of its own)
    mark_dominant(lib, ctx)
    lib.sayuri_hip_sync(ctx)
    t1 = time.perf_counter()
    if lib.sayuri_hip_time_runs(ctx, steps, ctypes.byref(ms)):
        raise RuntimeError(lib.sayuri_hip_last_error().decode())
    lib.sayuri_hip_sync(ctx)
    el_one = time.perf_counter() - t1
    stat = _lib.KernelStat()
    lib.sayuri_hip_timed_stat(ctx, ctypes.byref(stat))
    pipe.Destroy()
    px = float((bsz.astype(np.int64) ** 2).sum())
    flops_batch = sum(algorithmic_flops_per_eval(spec, int(b)) for b in bsz)
    tower_tf = (stat.flops / stat.launches) / (stat.total_ms / stat.launches * 1e-3) / 1e12 if stat.launches else None
    return {"workload": "configs[4]: 40-block x 384-filter net, fp16, batch 256 of mixed 9/13/19 boards (uniform draw, random order), "
                        "planes resident in HBM", "evals_per_sec": round(n * steps / el, 1), "ms_per_step": round(el / steps * 1e3, 3),
            "_evals": n * steps, "_seconds": el, "_flops": flops_batch * steps,
            "chains": chains, "ms_per_step_one_chain": round(el_one / steps * 1e3, 3), "evals_per_sec_one_chain": round(n * steps / el_one, 1),
            "real_pixel_fraction": round(px / (n * 361), 4), "gflop_per_batch": round(flops_batch / 1e9, 1),
            "whole_net_tflops": round(flops_batch * steps / el / 1e12, 1), "whole_net_mfma_frac": round(flops_batch * steps / el / 1e12 / 2500.0, 4),
            "tower_conv_avg_launch_us": round(stat.total_ms / max(stat.launches, 1) * 1e3, 2),
            "tower_conv_tflops": round(tower_tf, 1) if tower_tf else None, "tower_conv_mfma_frac": round(tower_tf / 2500.0, 4) if tower_tf else None}


def pump_segment(lib, ctx, grid, n: int, steps: int, packed: bool = False) -> dict:
    """The production form of the same step: what the pump thread of HipForwardPipe does (hip_forward_pipe.cc) -- two
    batches in flight on the engine's streams (copies in, forward, copies out) through sayuri_hip_submit[_packed] /
    sayuri_hip_wait, planes from pinned host buffers (H2D), outputs back into pinned buffers (D2H).  PCIe-inclusive, so it
    is reported beside `value`, never as `value`."""
    FP = ctypes.POINTER(ctypes.c_float)
    lib.sayuri_hip_host_alloc.restype = ctypes.c_void_p
    lib.sayuri_hip_host_alloc.argtypes = [ctypes.c_size_t]
    lib.sayuri_hip_host_free.argtypes = [ctypes.c_void_p]
    lib.sayuri_hip_submit.argtypes = [ctypes.c_void_p, ctypes.c_int, FP, ctypes.POINTER(ctypes.c_int), FP, FP, FP, FP,
                                      ctypes.POINTER(ctypes.c_int)]
    lib.sayuri_hip_wait.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.sayuri_hip_submit_packed.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                             FP, FP, FP, FP, ctypes.POINTER(ctypes.c_int)]
    B2 = grid.shape[2]
    if packed:  # the compact planes of csrc/host/packed_planes.h: 37 bit planes + 6 scalars, 1.8 KB per sample
        from sayuri_amd.engine import pack_planes
        inp = np.stack([pack_planes(grid[i], 37) for i in range(n)]).view(np.float32)
    else:
        inp = grid
    sizes = (inp.size, n * 5 * B2, n * 8, n * 32, n * B2)  # planes, prob, pass, misc (generous), own
    raw, bufs = [], []
    for _ in range(2):
        ptrs = [lib.sayuri_hip_host_alloc(k * 4) for k in sizes]
        if not all(ptrs):
            raise RuntimeError("sayuri_hip_host_alloc failed")
        raw += ptrs
        np.ctypeslib.as_array(ctypes.cast(ptrs[0], FP), (inp.size,))[:] = inp.ravel()
        bufs.append([ctypes.cast(q, FP) for q in ptrs])
    tick = [ctypes.c_int(-1), ctypes.c_int(-1)]

    def submit(i):
        pl, pr, pa, mi, ow = bufs[i]
        if packed:
            rc = lib.sayuri_hip_submit_packed(ctx, n, ctypes.cast(pl, ctypes.c_void_p), 37, None, pr, pa, mi, ow, ctypes.byref(tick[i]))
        else:
            rc = lib.sayuri_hip_submit(ctx, n, pl, None, pr, pa, mi, ow, ctypes.byref(tick[i]))
        if rc:
            raise RuntimeError(lib.sayuri_hip_last_error().decode())

    def wait(i):
        if lib.sayuri_hip_wait(ctx, tick[i].value):
            raise RuntimeError(lib.sayuri_hip_last_error().decode())

    for _ in range(2):
        submit(0); wait(0)
    steps = max(steps, 4)
    t0 = time.perf_counter()
    submit(0); submit(1)
    for k in range(steps - 2):
        wait(k & 1); submit(k & 1)
    wait(steps & 1); wait((steps + 1) & 1)
    dt = time.perf_counter() - t0
    for q in raw:
        lib.sayuri_hip_host_free(ctypes.c_void_p(q))
    return {"nn_evals_per_sec": round(n * steps / dt, 1), "ms_per_batch": round(dt / steps * 1e3, 4), "batches": steps, "in_flight": 2,
            "h2d_bytes_per_eval": int(inp.size * 4 // n),
            "what": ("sayuri_hip_submit_packed/wait: H2D of packed planes (37 bit planes + 6 scalars) from pinned memory + forward + D2H, "
                     if packed else
                     "sayuri_hip_submit/wait as the pump thread drives them: H2D of fp32 planes from pinned memory + forward + D2H, ") +
                    "two batches in flight (PCIe-inclusive, so not `value`)"}


def parity_sample(net, kind: str, pipe, planes, fp16: bool, k: int = 4) -> dict:
    """The benchmarked engine against the CPU reference on the first k positions of the bench's own batch -- so that the line
    that carries a throughput also carries the distance at which it was measured (north_star: 'within fp32 tolerance' holds for
    the fp32 engine, --fp32, 1e-4 abs; the default engine stores fp16 and accumulates fp32, like the reference's fp16 CUDA
    path, and is gated at 4e-3 x the output scale plus the reference's own SelfCheck, network.cc:333-359: L2 <= 0.2)."""
    from _oracle import PortNet
    got = pipe.BatchForward(planes[:k], [19] * k)
    err = scale = l2max = 0.0
    for p, g in zip(planes[:k], got):
        exp = net.forward(p, 19)
        err = max(err, float(np.abs(g - exp).max()))
        scale = max(scale, float(np.abs(exp).max()))
        a, b = PortNet.postprocess(g, 19), PortNet.postprocess(exp, 19)
        va = np.concatenate([a[:362], [a[2 * 361 + 1 + 3]]])
        vb = np.concatenate([b[:362], [b[2 * 361 + 1 + 3]]])
        l2max = max(l2max, float(np.sqrt(((va - vb) ** 2).sum())))
    gate = 4e-3 * max(1.0, scale) if fp16 else 1e-4
    return {"engine": "fp16 storage / MFMA with fp32 accumulation" if fp16 else "fp32 storage / fp32 MFMA",
            "against": f"the {kind} CPU pipe (BlasForwardPipe) on the first {k} positions of this run's batch, raw outputs",
            "max_abs_err": round(err, 6), "output_scale": round(scale, 3), "gate": round(gate, 6),
            "gate_rule": "4e-3 x max(1, output scale) and SelfCheck L2 <= 0.2 (fp16 engine)" if fp16 else "1e-4 abs (SURVEY 8c)",
            "selfcheck_l2_max": round(l2max, 6), "within_gate": bool(err <= gate and l2max <= 0.2),
            "strict_engine": "python bench.py --fp32: fp32 storage and MFMA, gated at 1e-4 abs, 5.7 k evals/s (profiles/r06_bench_fp32.json)"}


def cpu_baseline(weights_path: str, planes, seconds: float = 15.0, pipe=None, fp16: bool = True):
    """Time the CPU pipe on this box's host cores on a bounded sample of the same workload.
    Prefers the reference's own BlasForwardPipe (oracle/_ref, kind "reference"); falls back to
    the C restatement (kind "port").  One independent evaluation per thread at a time, like the
    reference's CPU mode (batch 1 per search thread, config.cc:254-265)."""
    from _oracle import PortNet, RefNet, ref_available
    threads = max(1, min(os.cpu_count() or 1, 64))
    if ref_available():
        net, kind = RefNet(weights_path, True), "reference"
    else:
        net, kind = PortNet(weights_path, True), "port"
    # single-thread calibration (also warms the caches)
    t0 = time.perf_counter()
    net.forward(planes[0], 19)
    one = time.perf_counter() - t0
    per_thread = max(1, int(round(seconds / max(one, 1e-3) / 1.5)))
    counts = [0] * threads

    def work(i):
        for k in range(per_thread):
            net.forward(planes[(i + k) % len(planes)], 19)
            counts[i] += 1

    ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    total = sum(counts)
    parity = parity_sample(net, kind, pipe, planes, fp16) if pipe is not None else None
    return {"parity": parity, "value": round(total / dt, 3), "unit": "evals/s", "cores": threads, "kind": kind,
            "sample": f"{total} evals of the same 20b256 19x19 net in {dt:.1f}s, {threads} threads x batch 1 "
                      f"(1 thread alone: {1.0 / one:.2f} evals/s); the reference's BlasForwardPipe with its BUILT_IN sgemm -- "
                      "Eigen / OpenBLAS are not in this image, the three CPU variants differ only in the GEMM call (blas.cc:16-166)"}


def launch_ranks(n: int, port: int) -> int:
    """`python bench.py --gpus N` without a launcher: be the launcher.  Checks that the box has N devices (the HIP library's
    own count; SAYURI_BENCH_SHARE_DEVICE is the one-GPU test hook that lets ranks share), then runs this file's own command line
    under `torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1 -- one rank per GPU, rank 0 prints the JSON line
    to this process's stdout -- and returns its exit code."""
    import subprocess
    from sayuri_amd import _lib
    have = int(_lib.hip().sayuri_hip_device_count())
    if have < n and not os.environ.get("SAYURI_BENCH_SHARE_DEVICE"):
        print(f"bench.py: --gpus {n} but this box has {have} HIP device(s)", file=sys.stderr)
        return 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")             # what torch.distributed.run would set, without its warning
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--fp32", action="store_true", help="strict-parity fp32 engine instead of fp16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-pump", dest="pump", action="store_false",
                    help="skip the pipelined segment (two batches in flight through submit/wait, PCIe-inclusive)")
    ap.add_argument("--profile", action="store_true", help="also print the per-kernel-class table to stderr")
    ap.add_argument("--selfplay-seconds", type=float, default=150.0,
                    help="length of the self-play window (configs[2]: 512 concurrent 19x19 games, 400 visits); 0 = skip")
    ap.add_argument("--selfplay-stagger", type=int, default=360,
                    help="the first game of worker g starts after g*N/games policy-sampled moves (0 = every game from move 0): "
                         "a window of minutes then sees games in every phase, as hours of self-play do, and games/hour can be counted")
    ap.add_argument("--config5", dest="config5", action="store_true", default=True, help="(the default)")
    ap.add_argument("--no-config5", dest="config5", action="store_false",
                    help="skip configs[4] (40-block x 384 net, batch 256 of mixed 9/13/19 boards, on every rank; ~20 s: the "
                         "generated weights are cached under /tmp)")
    ap.add_argument("--selfplay-games", type=int, default=512, help="concurrent self-play games per GPU")
    ap.add_argument("--selfplay-chunk-pool", type=int, default=16,
                    help="finished games the data writer holds back in its shuffle pool (0 = the reference's rule: as many as "
                         "there are concurrent games, which in a window of minutes puts all the writing behind the window)")
    ap.add_argument("--selfplay-visits", type=int, default=400)
    ap.add_argument("--force-dist", action="store_true", default=bool(os.environ.get("SAYURI_BENCH_FORCE_DIST")),
                    help="initialise torch.distributed even for one rank (world size 1): the barrier, the stats all-gather and the "
                         "periodic exchange of the self-play window then really go through RCCL on a one-GPU box (also "
                         "SAYURI_BENCH_FORCE_DIST=1)")
    ap.add_argument("--no-exchange", action="store_true", help="self-play window without the periodic exchange (A/B of its cost)")
    ap.add_argument("--dist-backend", default=os.environ.get("SAYURI_DIST_BACKEND", "nccl"),
                    help="torch.distributed backend of the multi-rank run: nccl (= RCCL, the default) or gloo (CPU tests on the "
                         "fake device; also SAYURI_DIST_BACKEND)")
    ap.add_argument("--master-port", type=int, default=int(os.environ.get("SAYURI_BENCH_MASTER_PORT", "29571")),
                    help="rendezvous port of the ranks a plain `python bench.py --gpus N` starts itself")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # A plain `python bench.py --gpus N` (no launcher around it): start the N ranks here -- one process per GPU, the
        # command line the docstring names, with this process as the launcher.  The reference does the in-process form of
        # this: one NNGraph per listed GPU and one worker thread each (cuda_forward_pipe.cc:85-116, batch_forward_pipe.cc:80-97).
        sys.exit(launch_ranks(args.gpus, args.master_port))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; the line's n_gpus "
                         "would not be what was asked for")

    import torch
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        if args.force_dist and "RANK" not in os.environ:  # a plain `python bench.py --force-dist`: a world of one
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        if args.dist_backend == "nccl":
            dev_index = local_rank % max(1, torch.cuda.device_count()) if os.environ.get("SAYURI_BENCH_SHARE_DEVICE") else local_rank
            torch.cuda.set_device(dev_index)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=args.dist_backend)

    from sayuri_amd import _lib
    from sayuri_amd import weights as W
    from sayuri_amd.pipe import HipForwardPipe

    lib = _lib.hip()
    if lib.sayuri_hip_device_count() < (1 if os.environ.get("SAYURI_BENCH_SHARE_DEVICE") else local_rank + 1):
        raise SystemExit(f"bench.py: rank {rank} needs HIP device {local_rank}, this box has {lib.sayuri_hip_device_count()}")
    if os.environ.get("SAYURI_BENCH_SHARE_DEVICE"):
        # TEST HOOK (tests/test_gpu_dropin.py): the ranks of a multi-rank launch share the devices that exist, so that this file's
        # multi-rank path (barrier, max-over-ranks timing, stats gather, exchange rounds) runs against the real runtime on a
        # one-GPU box.  The throughput of such a run means nothing.
        device = local_rank % max(1, lib.sayuri_hip_device_count())
    else:
        device = local_rank
    spec = W.spec_20b256()
    # a directory of its own: the self-play loop watches it for newer networks (reference ShouldHalt, engine.cc:63-90)
    wdir = f"/tmp/sayuri_bench_weights_{os.getuid()}"
    wpath = os.path.join(wdir, "net_20b256_seed22.bin")
    if local_rank == 0 and not os.path.exists(wpath):
        os.makedirs(wdir, exist_ok=True)
        W.write_weights(wpath + ".tmp", spec, seed=22)
        os.replace(wpath + ".tmp", wpath)
    if dist is not None:
        dist.barrier()
    else:
        while not os.path.exists(wpath):
            time.sleep(0.1)

    fp16 = not args.fp32
    n = args.batch
    pipe = HipForwardPipe(wpath, board_size=19, batch_size=n, fp16=fp16, device=device)
    ctx = pipe.ctx(0)
    planes = W.synthetic_planes(n, 19, seed=1000 + rank)
    grid = np.ascontiguousarray(np.stack(planes), np.float32)  # [n][43][361]
    bsz = np.full(n, 19, np.int32)
    fp = lambda a: a.ctypes.data_as(_lib.c_float_p)
    if lib.sayuri_hip_upload(ctx, n, fp(grid), bsz.ctypes.data_as(_lib.c_int_p)):
        raise RuntimeError(lib.sayuri_hip_last_error().decode())

    def sync_all():
        lib.sayuri_hip_sync(ctx)
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    ms = ctypes.c_float(0)
    if args.warmup > 0:
        lib.sayuri_hip_mark_kernel(ctx, b"")
        if lib.sayuri_hip_time_runs(ctx, args.warmup, ctypes.byref(ms)):
            raise RuntimeError(lib.sayuri_hip_last_error().decode())
    dominant = mark_dominant(lib, ctx)
    tower_state = ("persistent" if lib.sayuri_hip_tower_state(ctx) == 1 and dominant == "tower_run" else
                   "per-layer (fp32 engine)" if not fp16 else "per-layer fallback")

    # ---- timed region: exactly K steps between barrier+sync pairs
    sync_all()
    if dist is not None:
        dist.barrier()
    sync_all()
    t0 = time.perf_counter()
    if lib.sayuri_hip_time_runs(ctx, args.steps, ctypes.byref(ms)):
        raise RuntimeError(lib.sayuri_hip_last_error().decode())
    sync_all()
    if dist is not None:
        dist.barrier()
    sync_all()
    elapsed = time.perf_counter() - t0

    stat = _lib.KernelStat()
    lib.sayuri_hip_timed_stat(ctx, ctypes.byref(stat))
    pump = pump_segment(lib, ctx, grid, n, max(args.steps, 40)) if args.pump else None
    pump_packed = pump_segment(lib, ctx, grid, n, max(args.steps, 40), packed=True) if args.pump else None

    from sayuri_amd.shard import gather_stats
    if dist is not None:
        # one small stats record per rank (RCCL all-gather)
        stats = gather_stats({"games_done": 0, "nn_queries": n * args.steps, "nn_batches": args.steps,
                              "elapsed": elapsed})
        elapsed = stats["elapsed_max"]
        assert stats["nn_queries"] == world * n * args.steps

    # ---- second segment: the self-play loop on the same pipe (encoder + search + cache + batched queue + PCIe), one
    # engine per rank, games sharded by rank, no data-path collective
    selfplay = None
    if args.selfplay_seconds > 0:
        from sayuri_amd import search as S
        from sayuri_amd.shard import PeriodicGather
        sp_opts = dict(playouts=args.selfplay_visits, parallel_games=args.selfplay_games, num_games=1000000, seed=1000 + rank,
                       dirichlet_noise=1, dirichlet_epsilon=0.25, dirichlet_init=0.03, dirichlet_factor=361, first_pass_bonus=1,
                       random_moves_factor=0.1, komi_stddev=2.5, komi_big_stddev_prob=0.06, komi_big_stddev=12, lcb_reduction=0.0,
                       resign_playouts=80, resign_threshold=0.05, resign_discard_prob=0.9, early_symm_cache=1, cache_memory_mib=400,
                       selfplay_query=["bkp:19:7:1"], stagger_moves=args.selfplay_stagger, weights_dir=wdir, weights_file=wpath)
        # The data writer is part of the reference's games/hour (SaveChunk + gzip, pipe.cc:116-159,181-233): every finished game's
        # records go through StreamOut and gzip level 9 into tdata/ vdata/ sgf/ net_queries/ under a scratch directory, which is
        # removed afterwards.  The reference holds `parallel_games` finished games back in its shuffle pool; in a window of minutes
        # that would put all the writing behind the window, so the pool is cut to --selfplay-chunk-pool games here (the writer's
        # steady state is one chunk out per finished game either way; the 27-minute profile keeps the reference's pool).
        import shutil
        import tempfile
        sp_dir = tempfile.mkdtemp(prefix=f"sayuri_bench_selfplay_r{rank}_")
        sp_opts.update(target_directory=sp_dir, chunk_pool_games=args.selfplay_chunk_pool)
        if dist is not None:
            dist.barrier()
        # the path's only exchange: every 2 s each rank contributes its counters and its halt wish (newer weights seen);
        # RCCL all-gather of ~80 bytes per rank (sayuri_amd/shard.py)
        pg = PeriodicGather()
        batches0 = pipe.pump_times()["batches"]

        def local_record(st):
            return {"games_done": st["games_done"], "nn_queries": st["nn_queries"], "nn_batches": pipe.pump_times()["batches"] - batches0,
                    "cache_hits": st["cache_hits"], "moves": st["moves"], "playouts": st["playouts"], "records": st["records"],
                    "elapsed": st["elapsed"]}

        import resource
        ru0, pt0 = resource.getrusage(resource.RUSAGE_SELF), pipe.pump_times()
        st = S.selfplay(pipe, sp_opts, seconds=args.selfplay_seconds, name_suffix=f"-r{rank}",
                        on_stats=None if args.no_exchange else (lambda snap, halt: pg.tick(local_record(snap), halt=halt)),
                        stats_interval=2.0)
        ru1, pt1 = resource.getrusage(resource.RUSAGE_SELF), pipe.pump_times()
        tot = pg.drain(local_record(st))
        fin = gather_stats({"games_done": st["finished_moves"], "moves": st["prerolled_moves"], "elapsed": st["elapsed"]})
        # the writer's counters of all ranks, through the same record (the keys are only slots here)
        wr = gather_stats({"games_done": st["chunks_saved"], "nn_queries": st["chunks_saved_window"], "nn_batches": st["bytes_written"],
                           "cache_hits": st["text_bytes"], "moves": st["writer_cpu_seconds"] * 1e6,
                           "playouts": st["writer_cpu_seconds_window"] * 1e6, "records": st["writer_flush_seconds"] * 1e6,
                           "elapsed": st["elapsed"]})
        files_on_disk = sum(len(fs) for _, _, fs in os.walk(sp_dir))
        shutil.rmtree(sp_dir, ignore_errors=True)
        el = tot["elapsed_max"]
        games, finished_moves = int(tot["games_done"]), int(fin["games_done"])
        moves_per_sec = tot["moves"] / el
        mean_len = finished_moves / games if games else None
        selfplay = {"workload": "configs[2]: 19x19, 20b x 256 net, %d visits/move, %d concurrent games per GPU, Dirichlet noise, NN cache "
                                "400 MiB; %.0f s window; the first game of worker g starts after g*%d/%d policy-sampled moves so that the "
                                "window sees games in every phase (games still running at the end are not counted)"
                                % (args.selfplay_visits, args.selfplay_games, args.selfplay_seconds, args.selfplay_stagger, args.selfplay_games),
                    "seconds": round(el, 2), "nn_evals_per_sec": round(tot["nn_queries"] / el, 1),
                    "playouts_per_sec": round(tot["playouts"] / el, 1), "moves_per_sec": round(moves_per_sec, 2),
                    "games_done": games,
                    # the reference's definition: games finished / wall (played_games_ over the run, src/selfplay/pipe.cc:272-280),
                    # every one of them written out (chunks_saved).  The window's first generation starts at pre-rolled positions
                    # (--selfplay-stagger), so over a window of minutes this is an upper bound of the long-run rate; the 27-minute
                    # profiles (profiles/r05_selfplay_27min_*.json) measure it from the empty board
                    "games_per_hour": round(games / el * 3600, 1) if games else None,
                    "games_per_hour_definition": "finished games / wall over the window, chunks written inside it (reference "
                                                 "pipe.cc:272-280); the first generation of games starts at pre-rolled positions, see "
                                                 "games_per_hour_from_move_rate for the figure that does not depend on the pre-roll",
                    # the steady-state rate: searched moves per second / moves a finished game had (independent of where the
                    # pre-rolled openings put the games of the first generation)
                    "games_per_hour_from_move_rate": round(moves_per_sec * 3600 / mean_len, 1) if mean_len else None,
                    "mean_moves_per_finished_game": round(mean_len, 1) if mean_len else None,
                    "prerolled_moves": int(fin["moves"]),
                    # the data writer: one gzip'ed tdata + vdata chunk and one SGF line per finished game
                    "chunks_saved": int(wr["games_done"]), "chunks_saved_in_window": int(wr["nn_queries"]),
                    "chunk_pool_games": args.selfplay_chunk_pool,
                    "records_written": int(tot["records"]),
                    "bytes_written": int(wr["nn_batches"]), "record_text_bytes": int(wr["cache_hits"]),
                    "files_on_disk": files_on_disk,
                    "writer_cpu_seconds": round(wr["moves"] / 1e6, 3), "writer_cpu_seconds_in_window": round(wr["playouts"] / 1e6, 3),
                    "writer_cores_in_window": round(wr["playouts"] / 1e6 / el / world, 4),
                    "writer_flush_seconds_after_window": round(wr["records"] / 1e6 / world, 3),
                    "mean_batch": round(tot["nn_queries"] / max(tot["nn_batches"], 1), 1),
                    "exchange_rounds": pg.rounds, "halt_seen": bool(pg.any_halt),
                    # the path's one collective: issue -> landed per round, on the self-play loop's calling thread, beside the
                    # persistent tower launches of the two batches in flight (sayuri_amd/shard.py)
                    "exchange": pg.latency_summary(),
                    # this rank's process: CPU seconds of all its threads (game fibers' scheduler threads, pump, writer, the
                    # runtime's own) from the call to its return -- the pre-rolled openings and the writer's final flush
                    # included, so an upper bound -- over the window's seconds; and the pump thread's account per batch
                    "host_cpu_cores_busy": round(((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / max(st["elapsed"], 1e-9), 2),
                    "pump_us_per_batch": {k: round((pt1[k] - pt0[k]) / max(pt1["batches"] - pt0["batches"], 1))
                                          for k in ("wait_batch_us", "gpu_queue_empty_us", "fill_us", "forward_us") if k in pt1},
                    "frac_of_microbench_evals": None}

    # ---- third segment: BASELINE.json configs[4] (40b x 384, mixed 9/13/19 boards) on EVERY rank's GPU -- the configuration is
    # quoted on 8 GPUs: each rank runs its own batch (weak scaling, no collective in the path), the line carries the sum of the
    # ranks' evaluations over the slowest rank's time, and the per-rank rates beside it
    config5 = None
    if args.config5:
        if dist is not None:
            dist.barrier()
        config5 = config5_segment(lib, local_rank, rank, min(args.steps, 30), min(args.warmup, 5), device=device)
        ev, sec, fl = config5.pop("_evals"), config5.pop("_seconds"), config5.pop("_flops")
        if dist is not None:
            g5 = gather_stats({"nn_queries": ev, "elapsed": sec, "playouts": fl / 1e9})
            config5["n_gpus"] = world
            config5["per_rank_evals_per_sec"] = [round(r["nn_queries"] / r["elapsed"], 1) for r in g5["per_rank"]]
            config5["evals_per_sec"] = round(g5["nn_queries"] / g5["elapsed_max"], 1)
            config5["whole_net_tflops"] = round(g5["playouts"] * 1e9 / g5["elapsed_max"] / 1e12, 1)
            config5["whole_net_mfma_frac"] = round(g5["playouts"] * 1e9 / g5["elapsed_max"] / 1e12 / (2500.0 * world), 4)
            config5["what_is_per_rank"] = "chains, one-chain figures and the tower convolution's launch time are rank 0's"

    result = None
    if rank == 0:
        flops_eval = algorithmic_flops_per_eval(spec)
        evals = world * n * args.steps
        value = evals / elapsed
        dtype = "f16" if fp16 else "f32"
        peak = PEAK_TFLOPS[dtype]
        ach = None
        if stat.launches > 0 and stat.total_ms > 0:
            ach = (stat.flops / stat.launches) / (stat.total_ms / stat.launches * 1e-3) / 1e12
        result = {
            "metric": "nn_evals_per_sec", "value": round(value, 1), "unit": "evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            # which form of the tower ran: the persistent launch, or the per-layer fallback a build takes when tower_seam.py
            # rejects the compiler's assembly (another hipcc) -- 1-2 % slower and without the in-kernel SE unit
            "tower": tower_state, "hipcc": toolchain_line(),
            "config": {"workload": "configs[1]: 19x19, 20-block x 256-filter net (SE every 3rd block, heads 32ch, "
                                   "mish), batch=256 inference microbench, planes resident in HBM",
                       "batch_per_gpu": n, "global_batch": n * world, "board": 19, "parallelism": f"dp{world}",
                       "gflop_per_eval": round(flops_eval / 1e9, 3),
                       "whole_net_tflops": round(value * flops_eval / 1e12, 2),
                       "whole_net_mfma_frac": round(value * flops_eval / 1e12 / (peak * world), 4),
                       "device_ms_per_step": round(ms.value / args.steps, 4)},
            "roofline": {"bound": "mfma", "kernel": ("conv_tower_kernel<4> (tower_run: the input convolution and the 40 tower convolutions, SE units "
                                                     "included, as ONE persistent launch; one workgroup per board)" if dominant == "tower_run" else
                                                     "conv_board_kernel<4> (conv3x3_tower: 256->256 3x3, one workgroup per board)" if fp16 else
                                                     "conv_mfma_kernel<float> (conv3x3_tower: 256->256 3x3, v_mfma_f32_16x16x4_f32)"),
                         "achieved": round(ach, 2) if ach else None, "peak": peak, "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4) if ach else None, **hbm_traffic(fp16, dominant),
                         "launches_timed": int(stat.launches),
                         "avg_launch_us": round(stat.total_ms / max(stat.launches, 1) * 1e3, 2),
                         "flops_per_launch": stat.flops / max(stat.launches, 1)},
        }
        if pump is not None:
            result["config"]["pump"] = pump
            result["config"]["pump_packed"] = pump_packed
        if selfplay is not None:
            selfplay["frac_of_microbench_evals"] = round(selfplay["nn_evals_per_sec"] / value, 4)
            selfplay.update(long_run_games_per_hour())
            result["selfplay"] = selfplay
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(wpath, planes, args.cpu_seconds, pipe=pipe, fp16=fp16)
            result["parity"] = result["cpu_baseline"].pop("parity")
        if config5 is not None:
            result["config5"] = config5
        if args.profile:
            rows = (_lib.KernelStat * 32)()
            k = lib.sayuri_hip_profile_run(ctx, rows, 32)
            tot = sum(rows[i].total_ms for i in range(k))
            print(f"{'kernel class':<18}{'launches':>9}{'ms':>10}{'%':>7}{'TFLOP/s':>10}{'GB/s':>9}", file=sys.stderr)
            for i in range(k):
                r = rows[i]
                tf = r.flops / (r.total_ms * 1e-3) / 1e12 if r.total_ms > 0 else 0
                gb = r.bytes / (r.total_ms * 1e-3) / 1e9 if r.total_ms > 0 else 0
                print(f"{r.name.decode():<18}{r.launches:>9}{r.total_ms:>10.3f}{100 * r.total_ms / tot:>7.1f}"
                      f"{tf:>10.1f}{gb:>9.0f}", file=sys.stderr)
            print(f"{'sum (serialised)':<18}{'':>9}{tot:>10.3f}", file=sys.stderr)
    pipe.Destroy()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
