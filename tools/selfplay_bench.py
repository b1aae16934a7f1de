#!/usr/bin/env python3
"""Self-play throughput on one GPU: BASELINE.json configs[2] (19x19, 20b x 256 net, 400 visits, 512 concurrent games).

    python tools/selfplay_bench.py [--seconds 120] [--games 512] [--playouts 400] [--net 20b256|6b96] [--board 19]
                                   [--num-games N] [--move-cap M] [--fp32]

With --seconds the run is a time window (games in progress at the end are dropped); with --num-games it plays
that many complete games.  Prints one JSON line with the counters and the derived rates.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=0.0)
    ap.add_argument("--games", type=int, default=512, help="concurrent games")
    ap.add_argument("--num-games", type=int, default=0, help="complete games to play (default: one per worker; with --seconds: no limit, a worker whose game ends starts the next)")
    ap.add_argument("--playouts", type=int, default=400)
    ap.add_argument("--net", default="20b256")
    ap.add_argument("--board", type=int, default=19)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--move-cap", type=int, default=0)
    ap.add_argument("--waittime", type=int, default=2)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--out", default="", help="target directory of the chunks (tdata/ vdata/ sgf/ net_queries/); default: a scratch "
                                              "directory under /tmp that is removed afterwards")
    ap.add_argument("--no-writer", action="store_true", help="no target directory: finished games are counted and dropped (the writer's "
                                                             "SaveChunk + gzip is NOT in the measurement then)")
    ap.add_argument("--chunk-pool", type=int, default=0, help="finished games held back in the writer's shuffle pool (0 = the reference's "
                                                              "rule: as many as there are concurrent games)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--game-threads", type=int, default=0, help="0 = one OS thread per game up to 1024 games, fibers above; N = fibers on N threads; -1 = threads")
    ap.add_argument("--stagger", type=int, default=0, help="game g of the first generation starts after g * N / games policy-sampled moves (0: all from the empty board)")
    ap.add_argument("--boards", default="", help="comma list of board sizes drawn uniformly per game (mixed-size batches), e.g. 9,13,19")
    args = ap.parse_args()

    from sayuri_amd import search as S
    from sayuri_amd import weights as W
    from sayuri_amd.pipe import HipForwardPipe

    spec = {"20b256": W.spec_20b256, "6b96": W.spec_6b96, "40b384": W.spec_40b384}[args.net]()
    wpath = f"/tmp/sayuri_selfplay_{args.net}_{os.getuid()}.bin"
    if not os.path.exists(wpath):
        W.write_weights(wpath, spec, seed=22)
    import shutil
    import tempfile
    scratch = None
    if not args.out and not args.no_writer:
        scratch = args.out = tempfile.mkdtemp(prefix="sayuri_selfplay_chunks_")
    pipe = HipForwardPipe(wpath, board_size=args.board, batch_size=args.batch, fp16=not args.fp32, waittime_ms=args.waittime)
    opts = dict(playouts=args.playouts, parallel_games=args.games, num_games=max(args.num_games, args.games) if (args.num_games or args.seconds <= 0) else 1000000, seed=args.seed,
                dirichlet_noise=1, dirichlet_epsilon=0.25, dirichlet_init=0.03, dirichlet_factor=361, first_pass_bonus=1,
                random_moves_factor=0.1, komi_stddev=2.5, komi_big_stddev_prob=0.06, komi_big_stddev=12, lcb_reduction=0.0,
                resign_playouts=80, resign_threshold=0.05, resign_discard_prob=0.9, early_symm_cache=1, cache_memory_mib=400,
                selfplay_query=([f"bkp:{b}:7:1" for b in args.boards.split(",")] if args.boards else [f"bkp:{args.board}:7:1"]),
                target_directory=args.out, game_threads=args.game_threads, stagger_moves=args.stagger, chunk_pool_games=args.chunk_pool)
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.time()
    series = []
    st = S.selfplay(pipe, opts, seconds=args.seconds, move_cap=args.move_cap,
                    on_stats=lambda snap, halt: series.append((snap['elapsed'], snap['nn_queries'], snap['moves'], snap['playouts'])) and False,
                    stats_interval=5.0)
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    pt = pipe.pump_times()
    el = st["elapsed"]
    out = dict(st)
    if args.out:
        out["files_on_disk"] = sum(len(fs) for _, _, fs in os.walk(args.out))
        out["writer_cores_in_window"] = round(st["writer_cpu_seconds_window"] / el, 4)
    if scratch:
        shutil.rmtree(scratch, ignore_errors=True)
    out.update(net=args.net, board=args.board, concurrent_games=args.games, playouts_per_move=args.playouts,
               nn_evals_per_sec=round(st["nn_queries"] / el, 1), playouts_per_sec=round(st["playouts"] / el, 1),
               moves_per_sec=round(st["moves"] / el, 2), games_per_hour=round(st["games_done"] / el * 3600, 1),
               cache_hit_rate=round(st["cache_hits"] / max(st["cache_lookups"], 1), 4),
               mean_batch=round(pt["evals"] / max(pt["batches"], 1), 1), partial_batches=pt["partial_batches"], batches=pt["batches"],
               pump_us_per_batch={k: round(pt[k] / max(pt["batches"], 1)) for k in ("forward_us", "fill_us", "wait_batch_us", "wait_copies_us", "wake_parked_us", "gpu_queue_empty_us", "wait_plane_copies_us")},
               host_cpu_cores_busy=round(((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / el, 1),
               host_sys_cores=round((ru1.ru_stime - ru0.ru_stime) / el, 1), host_cpus=os.cpu_count(),
               ctx_switches_per_sec=round(((ru1.ru_nvcsw - ru0.ru_nvcsw) + (ru1.ru_nivcsw - ru0.ru_nivcsw)) / el),
               max_rss_gb=round(ru1.ru_maxrss / 1048576, 2), wall=round(time.time() - t0, 1))
    if len(series) >= 4:  # the rate once the start-up transient (first trees, first-touch page faults) is over
        a, b = series[len(series) // 2], series[-1]
        dt = max(b[0] - a[0], 1e-9)
        out.update(second_half={"from_s": round(a[0], 1), "to_s": round(b[0], 1), "nn_evals_per_sec": round((b[1] - a[1]) / dt, 1),
                                "moves_per_sec": round((b[2] - a[2]) / dt, 2), "playouts_per_sec": round((b[3] - a[3]) / dt, 1)})
    print(json.dumps(out))
    pipe.Destroy()


if __name__ == "__main__":
    main()
