#!/usr/bin/env python3
"""Multi-rank readiness without the 8-GPU node: the whole host side of the sharded self-play run --
one process per "GPU", each with its own queue, games (fibers), NN cache and chunk directory, the periodic all-gather of
the stats record and the collective ShouldHalt -- on tests/fake_hip (a SERIAL device that takes FAKE_HIP_SERIAL_US per
batch, 3 800 us = one 256-batch of the 20b x 256 network on an MI355X, with a network that costs the host nothing).

    python tools/fake8.py --ranks 8 --games 512 --seconds 60 --out profiles/r03_fake8_ranks.json
    python tools/fake8.py --ranks 1 --devices 8 --games 4096 --seconds 60 --out profiles/r03_fake8_inprocess.json

Per rank it records NN evals/s, host core-seconds per evaluation (getrusage user + system over the run), resident set and
the exchange rounds; rank `--halt-rank` drops a newer network into its weights directory after `--halt-after` seconds and
every rank must wind down (reference Engine::ShouldHalt, engine.cc:63-90; pipe.cc:246-258).  Backend of the exchange: gloo.
The container this runs in has few cores: what carries over to the node is core-us per evaluation and bytes per rank, not
the rate."""
from __future__ import annotations

import argparse
import json
import os
import resource
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FAKE_SRC = os.path.join(ROOT, "tests", "fake_hip", "fake_hip.c")


def child(args):
    import ctypes
    ctypes.CDLL(args.fake, mode=ctypes.RTLD_GLOBAL)
    import torch.distributed as dist
    from sayuri_amd import search as S
    from sayuri_amd.pipe import HipForwardPipe
    from sayuri_amd.shard import PeriodicGather, gather_stats
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if world > 1:
        dist.init_process_group(backend="gloo")
    wdir = os.path.join(args.work, f"weights-r{rank}")
    os.makedirs(wdir, exist_ok=True)
    wpath = os.path.join(wdir, "net0.bin")
    shutil.copy(args.weights, wpath)
    pipe = HipForwardPipe(wpath, board_size=19, batch_size=256, fp16=True, device=-1 if args.devices > 1 else 0, waittime_ms=2)
    opts = dict(playouts=args.playouts, parallel_games=args.games, num_games=1000000, seed=1000 + rank, dirichlet_noise=1,
                dirichlet_epsilon=0.25, dirichlet_init=0.03, dirichlet_factor=361, first_pass_bonus=1, random_moves_factor=0.1,
                komi_stddev=2.5, komi_big_stddev_prob=0.06, komi_big_stddev=12, lcb_reduction=0.0, resign_playouts=80,
                resign_threshold=0.05, resign_discard_prob=0.9, early_symm_cache=1, cache_memory_mib=400,
                selfplay_query=[f"bkp:{args.board}:7:1"], weights_dir=wdir + "/", weights_file=wpath,  # another spelling of the directory:
                # the halt check compares file identity (st_dev, st_ino), not strings -- a run must not halt over its own network
                target_directory=os.path.join(args.work, f"out-r{rank}"))
    os.makedirs(opts["target_directory"], exist_ok=True)
    pg = PeriodicGather()
    dropped = [False]
    rss_series = []

    def on_stats(snap, halt):
        if rank == args.halt_rank and not dropped[0] and args.halt_after > 0 and snap["elapsed"] >= args.halt_after:
            shutil.copy(args.weights, os.path.join(wdir, "net1.bin"))  # a newer network appears on ONE rank
            dropped[0] = True
        rss_series.append((round(snap["elapsed"], 1), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024))
        rec = {"games_done": snap["games_done"], "nn_queries": snap["nn_queries"], "moves": snap["moves"],
               "playouts": snap["playouts"], "elapsed": snap["elapsed"]}
        return pg.tick(rec, halt=halt) if world > 1 else halt

    ru0, t0 = resource.getrusage(resource.RUSAGE_SELF), time.time()
    st = S.selfplay(pipe, opts, seconds=args.seconds, name_suffix=f"-r{rank}", on_stats=on_stats, stats_interval=2.0)
    ru1, wall = resource.getrusage(resource.RUSAGE_SELF), time.time() - t0
    if world > 1:
        pg.drain({"games_done": st["games_done"], "nn_queries": st["nn_queries"], "elapsed": st["elapsed"]})
    pt = pipe.pump_times()
    cpu = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    out = {"rank": rank, "elapsed": round(st["elapsed"], 2), "nn_evals": st["nn_queries"],
           "nn_evals_per_sec": round(st["nn_queries"] / st["elapsed"], 1), "mean_batch": round(pt["evals"] / max(pt["batches"], 1), 1),
           "batches": pt["batches"], "moves": st["moves"], "games_done": st["games_done"], "playouts": st["playouts"],
           "cpu_seconds": round(cpu, 2), "system_seconds": round(ru1.ru_stime - ru0.ru_stime, 2),
           "cores_busy": round(cpu / wall, 2), "core_us_per_eval": round(cpu / max(st["nn_queries"], 1) * 1e6, 1),
           "max_rss_mb": ru1.ru_maxrss // 1024, "rss_mb_over_time": rss_series[:: max(1, len(rss_series) // 12)],
           "exchange_rounds": pg.rounds, "halt_seen": bool(pg.any_halt) if world > 1 else dropped[0],
           "stopped_early_by_halt": st["elapsed"] < args.seconds - 3}
    pipe.Destroy()
    json.dump(out, open(os.path.join(args.work, f"rank{rank}.json"), "w"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--devices", type=int, default=1, help="fake devices per process (in-process multi-GPU form of the drop-in)")
    ap.add_argument("--games", type=int, default=512, help="concurrent games per rank")
    ap.add_argument("--playouts", type=int, default=400)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--board", type=int, default=19, help="board size of the games (the halt check runs once per game of worker 0: small boards make it testable in seconds)")
    ap.add_argument("--serial-us", type=int, default=3800)
    ap.add_argument("--halt-rank", type=int, default=3)
    ap.add_argument("--halt-after", type=float, default=0.0, help="> 0: that rank sees a newer network after this many seconds")
    ap.add_argument("--out", default="")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--fake", default="")
    ap.add_argument("--work", default="")
    ap.add_argument("--weights", default="")
    args = ap.parse_args()
    if args.child:
        return child(args)

    from sayuri_amd import weights as W
    work = tempfile.mkdtemp(prefix="fake8_")
    fake = os.path.join(work, "libfake_hip.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", FAKE_SRC, "-o", fake, "-lpthread"])
    weights = os.path.join(work, "net_6b96.bin")  # the fake device ignores the tensors; the loader and the heads' shapes are real
    W.write_weights(weights, W.spec_6b96(), seed=3)
    port = 29500 + os.getpid() % 2000
    procs = []
    t0 = time.time()
    for r in range(args.ranks):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(args.ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FAKE_HIP_SERIAL_US=str(args.serial_us), FAKE_HIP_CHEAP="1", FAKE_HIP_DEVICES=str(args.devices),
                   PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--fake", fake, "--work", work, "--weights", weights,
               "--ranks", str(args.ranks), "--devices", str(args.devices), "--games", str(args.games), "--playouts", str(args.playouts),
               "--seconds", str(args.seconds), "--board", str(args.board), "--halt-rank", str(args.halt_rank), "--halt-after", str(args.halt_after)]
        procs.append(subprocess.Popen(cmd, env=env, cwd=ROOT))
    rcs = [p.wait(timeout=args.seconds * 4 + 600) for p in procs]
    wall = time.time() - t0
    ranks = [json.load(open(os.path.join(work, f"rank{r}.json"))) for r in range(args.ranks) if os.path.exists(os.path.join(work, f"rank{r}.json"))]
    evals = sum(r["nn_evals"] for r in ranks)
    cpu = sum(r["cpu_seconds"] for r in ranks)
    pace = 256 / (args.serial_us * 1e-6) * args.devices  # evals/s one rank's device(s) can take
    summary = {
        "what": f"{args.ranks} process(es) x {args.devices} fake device(s) x {args.games} fiber games, {args.playouts} visits, {args.board}x{args.board}, "
                f"serial device {args.serial_us} us per batch (= {pace:.0f} evals/s per rank when fed), gloo exchange every 2 s; "
                f"host: {os.cpu_count()} cores visible in this container",
        "return_codes": rcs, "wall_seconds": round(wall, 1), "ranks_reporting": len(ranks),
        "nn_evals_per_sec_all": round(sum(r["nn_evals_per_sec"] for r in ranks), 1),
        "core_us_per_eval": round(cpu / max(evals, 1) * 1e6, 1),
        "cores_per_rank_at_device_pace": round(cpu / max(evals, 1) * pace, 2),
        "max_rss_mb_per_rank": max((r["max_rss_mb"] for r in ranks), default=0),
        "sum_rss_mb": sum(r["max_rss_mb"] for r in ranks),
        "exchange_rounds": [r["exchange_rounds"] for r in ranks],
        "halt_seen_by": [r["rank"] for r in ranks if r["halt_seen"]],
        "stopped_early_by_halt": [r["rank"] for r in ranks if r["stopped_early_by_halt"]],
        "per_rank": ranks,
    }
    text = json.dumps(summary, indent=1)
    if args.out:
        open(args.out, "w").write(text + "\n")
    print(json.dumps({k: v for k, v in summary.items() if k != "per_rank"}, indent=1))
    shutil.rmtree(work, ignore_errors=True)
    return 0 if all(rc == 0 for rc in rcs) and len(ranks) == args.ranks else 1


if __name__ == "__main__":
    sys.exit(main())
