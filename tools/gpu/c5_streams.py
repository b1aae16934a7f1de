#!/usr/bin/env python3
"""configs[4] (40b x 384, batch 256 of mixed 9/13/19 boards): does the hardware fill the second round of a layer when the
batch is run as G independent chains on G streams?

A layer of the whole batch is 450 workgroups of equal cost on 256 CUs: two rounds, the second 76 % full, whatever the item
size (profiles/r04_config5_analysis.txt).  The boards of a batch are independent, so the batch can be cut into G groups of
samples, each a chain of per-layer launches on its own stream (150 workgroups per launch at G = 3): a group's next layer then
starts on the CUs another group's round leaves free -- the cross-layer pipelining of a persistent (layer, tile, channel tile)
run, done by the dispatcher instead of by dependency counters.  This script measures it from outside the engine: G contexts
on one device, each with 1/G of the batch (same mix of sizes), timed concurrently from G host threads.

    python tools/gpu/c5_streams.py [--groups 1,2,3,4] [--steps 30]
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from sayuri_amd import _lib  # noqa: E402
from sayuri_amd import weights as W  # noqa: E402
from sayuri_amd.pipe import HipForwardPipe  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", default="1,2,3,4,1")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--net", default="40b384")
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--full", action="store_true", help="every group evaluates the WHOLE batch (G batches in flight on G streams: does a "
                                                        "second batch hide the first one's launch boundaries and tails?) instead of 1/G of it")
    ap.add_argument("--uniform", action="store_true", help="19x19 boards only (configs[1]'s batch) instead of the 9/13/19 mix")
    args = ap.parse_args()
    lib = _lib.hip()
    spec = {"40b384": W.spec_40b384, "20b256": W.spec_20b256}[args.net]()
    wpath = f"/tmp/sayuri_c5_{args.net}.bin"
    if not os.path.exists(wpath):
        W.write_weights(wpath, spec, seed=23)
    n = args.n
    rng = np.random.default_rng(5000)
    bsz = rng.choice([9, 13, 19], size=n).astype(np.int32)
    if args.uniform:
        bsz[:] = 19
    planes = W.synthetic_planes(n, [int(b) for b in bsz], seed=5100)
    grid = np.zeros((n, 43, 19, 19), np.float32)
    for i, (p, b) in enumerate(zip(planes, bsz)):
        grid[i, :, :b, :b] = p.reshape(43, b, b)
    grid = np.ascontiguousarray(grid.reshape(n, 43, 361))
    order = np.argsort(-bsz, kind="stable")  # deal the samples out by size so that every group gets the same mix
    out = []
    for G in [int(x) for x in args.groups.split(",")]:
        pipes, ctxs, counts = [], [], []
        for g in range(G):
            idx = order if args.full else order[g::G]
            pipe = HipForwardPipe(wpath, board_size=19, batch_size=len(idx), fp16=True)
            ctx = pipe.ctx(0)
            gg = np.ascontiguousarray(grid[idx])
            bb = np.ascontiguousarray(bsz[idx])
            if lib.sayuri_hip_upload(ctx, len(idx), gg.ctypes.data_as(_lib.c_float_p), bb.ctypes.data_as(_lib.c_int_p)):
                raise RuntimeError(lib.sayuri_hip_last_error().decode())
            pipes.append(pipe); ctxs.append(ctx); counts.append(len(idx))
        ms = [ctypes.c_float(0) for _ in range(G)]

        def run(g, iters):
            lib.sayuri_hip_mark_kernel(ctxs[g], b"")
            if lib.sayuri_hip_time_runs(ctxs[g], iters, ctypes.byref(ms[g])):
                raise RuntimeError(lib.sayuri_hip_last_error().decode())

        for iters in (5, args.steps):
            ths = [threading.Thread(target=run, args=(g, iters)) for g in range(G)]
            for c in ctxs:
                lib.sayuri_hip_sync(c)
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            for c in ctxs:
                lib.sayuri_hip_sync(c)
            el = time.perf_counter() - t0
        row = {"groups": G, "samples_per_group": counts, "ms_per_batch": round(el / args.steps * 1e3, 3),
               "evals_per_sec": round((G if args.full else 1) * n * args.steps / el, 1), "full_batch_per_group": bool(args.full), "device_ms_per_group_forward": [round(m.value / args.steps, 3) for m in ms]}
        print(json.dumps(row), flush=True)
        out.append(row)
        for p in pipes:
            p.Destroy()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"c5_streams_{args.net}{'_full' if args.full else ''}.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
