#!/usr/bin/env python3
"""configs[4] through the queue's path: sayuri_hip_submit_packed / wait with two tickets in flight (what the pump thread of
HipForwardPipe does), a mixed 9/13/19 batch of 256 on the 40b x 384 network, under the engine's stream switches:
SAYURI_CHAINS (1 = one chain per forward, 0 = the engine's choice) x SAYURI_COMPUTE_STREAMS (1 = both tickets on one compute
stream, 2 = a stream per ticket).  Prints one JSON line per setting; each setting twice, interleaved.

    python tools/gpu/c5_pump.py [--steps 40]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from sayuri_amd import _lib  # noqa: E402
from sayuri_amd import weights as W  # noqa: E402
from sayuri_amd.engine import pack_planes  # noqa: E402
from sayuri_amd.pipe import HipForwardPipe  # noqa: E402


def run(lib, wpath, records, bsz, n, steps):
    FP = ctypes.POINTER(ctypes.c_float)
    lib.sayuri_hip_host_alloc.restype = ctypes.c_void_p
    lib.sayuri_hip_host_alloc.argtypes = [ctypes.c_size_t]
    lib.sayuri_hip_host_free.argtypes = [ctypes.c_void_p]
    lib.sayuri_hip_wait.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.sayuri_hip_submit_packed.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                             FP, FP, FP, FP, ctypes.POINTER(ctypes.c_int)]
    pipe = HipForwardPipe(wpath, board_size=19, batch_size=n, fp16=True)
    ctx = pipe.ctx(0)
    B2 = 361
    sizes = (records.size, n * 5 * B2, n * 8, n * 32, n * B2)
    raw, bufs = [], []
    for _ in range(2):
        ptrs = [lib.sayuri_hip_host_alloc(k * 4) for k in sizes]
        raw += ptrs
        np.ctypeslib.as_array(ctypes.cast(ptrs[0], ctypes.POINTER(ctypes.c_uint32)), (records.size,))[:] = records.ravel()
        bufs.append(ptrs)
    hb = lib.sayuri_hip_host_alloc(n * 4)
    np.ctypeslib.as_array(ctypes.cast(hb, ctypes.POINTER(ctypes.c_int32)), (n,))[:] = bsz
    bp = ctypes.cast(hb, ctypes.POINTER(ctypes.c_int))
    tick = [ctypes.c_int(-1), ctypes.c_int(-1)]

    def submit(i):
        pl, pr, pa, mi, ow = bufs[i]
        if lib.sayuri_hip_submit_packed(ctx, n, ctypes.c_void_p(pl), 37, bp, ctypes.cast(pr, FP), ctypes.cast(pa, FP), ctypes.cast(mi, FP),
                                        ctypes.cast(ow, FP), ctypes.byref(tick[i])):
            raise RuntimeError(lib.sayuri_hip_last_error().decode())

    def wait(i):
        if lib.sayuri_hip_wait(ctx, tick[i].value):
            raise RuntimeError(lib.sayuri_hip_last_error().decode())

    for _ in range(3):
        submit(0); wait(0)
    t0 = time.perf_counter()
    submit(0); submit(1)
    for k in range(steps - 2):
        wait(k & 1); submit(k & 1)
    wait(steps & 1); wait((steps + 1) & 1)
    dt = time.perf_counter() - t0
    chains = int(lib.sayuri_hip_last_chains(ctx))
    for q in raw + [hb]:
        lib.sayuri_hip_host_free(ctypes.c_void_p(q))
    pipe.Destroy()
    return n * steps / dt, dt / steps * 1e3, chains


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--zc-in", action="store_true", help="SAYURI_IO_ZC_IN=1 against 0 under the engine's own stream choices")
    args = ap.parse_args()
    lib = _lib.hip()
    wpath = "/tmp/sayuri_c5_40b384.bin"
    if not os.path.exists(wpath):
        W.write_weights(wpath, W.spec_40b384(), seed=23)
    n = 256
    rng = np.random.default_rng(5000)
    bsz = rng.choice([9, 13, 19], size=n).astype(np.int32)
    planes = W.synthetic_planes(n, [int(b) for b in bsz], seed=5100)
    records = np.stack([pack_planes(p, 37) for p in planes])
    out = []
    settings = [("1", "1", "1"), ("1", "2", "1"), ("0", "1", "1"), ("0", "2", "1")]
    if args.zc_in:  # the engine's own choices, packed records read in place (default) against copied first
        settings = [("0", "1", "1"), ("0", "1", "0")]
    for rep in range(2):
        for chains, streams, zc_in in settings:
            os.environ["SAYURI_CHAINS"] = chains
            os.environ["SAYURI_COMPUTE_STREAMS"] = streams
            os.environ["SAYURI_IO_ZC_IN"] = zc_in
            eps, ms, got = run(lib, wpath, records, bsz, n, args.steps)
            row = {"SAYURI_CHAINS": chains, "SAYURI_COMPUTE_STREAMS": streams, "SAYURI_IO_ZC_IN": zc_in, "chains_per_forward": got,
                   "evals_per_sec": round(eps, 1), "ms_per_batch": round(ms, 3), "rep": rep}
            print(json.dumps(row), flush=True)
            out.append(row)
    with open(os.path.join(ROOT, "gpurun_out", "c5_pump.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
