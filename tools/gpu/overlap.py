"""Two half-batches in flight on the engine's two compute streams vs one full batch at a time:
does de-synchronising the workgroups (epilogue of one half under the K loop of the other) pay?"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sayuri_amd import _lib, weights as W  # noqa: E402
from sayuri_amd.pipe import HipForwardPipe  # noqa: E402

lib = _lib.hip()
lib.sayuri_hip_host_alloc.restype = ctypes.c_void_p
lib.sayuri_hip_host_alloc.argtypes = [ctypes.c_size_t]
FP = ctypes.POINTER(ctypes.c_float)
lib.sayuri_hip_submit.argtypes = [ctypes.c_void_p, ctypes.c_int, FP, ctypes.POINTER(ctypes.c_int), FP, FP, FP, FP, ctypes.POINTER(ctypes.c_int)]
lib.sayuri_hip_wait.argtypes = [ctypes.c_void_p, ctypes.c_int]
d = "/tmp/overlap_w"
os.makedirs(d, exist_ok=True)
path = os.path.join(d, "w.bin")
if not os.path.exists(path):
    W.write_weights(path, W.spec_20b256(), seed=1)
pipe = HipForwardPipe(path, board_size=19, batch_size=256, fp16=True)
ctx = ctypes.c_void_p(pipe.ctx(0))
B2 = 361


def pinned(n_floats):
    p = lib.sayuri_hip_host_alloc(n_floats * 4)
    return ctypes.cast(p, ctypes.POINTER(ctypes.c_float))


def run(n, inflight, iters):
    bufs = []
    for _ in range(2):
        planes = pinned(n * 43 * B2)
        np.ctypeslib.as_array(planes, (n * 43 * B2,))[:] = np.random.default_rng(1).random(n * 43 * B2, dtype=np.float32) > 0.5
        bufs.append((planes, pinned(n * 5 * B2), pinned(n * 8), pinned(n * 32), pinned(n * B2)))
    tick = [ctypes.c_int(-1), ctypes.c_int(-1)]
    def submit(i):
        pl, pr, pa, mi, ow = bufs[i]
        rc = lib.sayuri_hip_submit(ctx, n, pl, None, pr, pa, mi, ow, ctypes.byref(tick[i]))
        assert rc == 0, lib.sayuri_hip_last_error()
    for _ in range(3):
        submit(0); lib.sayuri_hip_wait(ctx, tick[0].value)
    t0 = time.perf_counter()
    if inflight == 1:
        for _ in range(iters):
            submit(0); lib.sayuri_hip_wait(ctx, tick[0].value)
    else:
        submit(0); submit(1)
        for k in range(iters - 2):
            i = k & 1
            lib.sayuri_hip_wait(ctx, tick[i].value); submit(i)
        lib.sayuri_hip_wait(ctx, tick[0].value); lib.sayuri_hip_wait(ctx, tick[1].value)
    dt = time.perf_counter() - t0
    print(f"batch {n:4d} x {inflight} in flight: {n * iters / dt:9.0f} evals/s  ({dt / iters * 1e3:.3f} ms per batch)", flush=True)


for n, f, it in ((256, 1, 200), (256, 2, 200), (128, 1, 400), (128, 2, 400), (64, 2, 800)):
    run(n, f, it)
pipe.Destroy()
