#!/usr/bin/env python3
"""Per-kernel-class device time of ONE forward of configs[4]'s batch (40b x 384, 256 mixed 9/13/19 boards), events around every
launch (sayuri_hip_profile_run): where the SE unit's time goes with SAYURI_SE_SPLIT=1 / 0.   python tools/gpu/c5_profile.py [--uniform]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from sayuri_amd import _lib, weights as W
from sayuri_amd.pipe import HipForwardPipe

lib = _lib.hip()
spec = W.spec_40b384()
wpath = f"/tmp/sayuri_bench_weights_{os.getuid()}_c5/net_40b384_seed23.bin"
if not os.path.exists(wpath):
    os.makedirs(os.path.dirname(wpath), exist_ok=True)
    W.write_weights(wpath, spec, seed=23)
n = 256
rng = np.random.default_rng(5000)
bsz = np.full(n, 19, np.int32) if "--uniform" in sys.argv else rng.choice([9, 13, 19], size=n).astype(np.int32)
planes = W.synthetic_planes(n, [int(b) for b in bsz], seed=5100)
grid = np.zeros((n, 43, 19, 19), np.float32)
for i, (p, b) in enumerate(zip(planes, bsz)):
    grid[i, :, :b, :b] = p.reshape(43, b, b)
grid = np.ascontiguousarray(grid.reshape(n, 43, 361))
pipe = HipForwardPipe(wpath, board_size=19, batch_size=n, fp16=True)
ctx = pipe.ctx(0)
assert lib.sayuri_hip_upload(ctx, n, grid.ctypes.data_as(_lib.c_float_p), bsz.ctypes.data_as(_lib.c_int_p)) == 0
ms = ctypes.c_float(0)
lib.sayuri_hip_mark_kernel(ctx, b"")
lib.sayuri_hip_time_runs(ctx, 5, ctypes.byref(ms))
import time
lib.sayuri_hip_time_runs(ctx, 10, ctypes.byref(ms)); lib.sayuri_hip_sync(ctx)
t0 = time.perf_counter(); lib.sayuri_hip_time_runs(ctx, 40, ctypes.byref(ms)); lib.sayuri_hip_sync(ctx); dt = time.perf_counter() - t0
print(f"forwards as the engine runs them: {n * 40 / dt:.0f} evals/s, {dt / 40 * 1e3:.3f} ms per batch, chains {lib.sayuri_hip_last_chains(ctx)}")
for rep in range(2):
    rows = (_lib.KernelStat * 32)()
    k = lib.sayuri_hip_profile_run(ctx, rows, 32)
tot = sum(rows[i].total_ms for i in range(k))
print(f"SAYURI_SE_SPLIT={os.environ.get('SAYURI_SE_SPLIT', '(default 1)')}  {'uniform 19x19' if '--uniform' in sys.argv else 'mixed 9/13/19'}")
for i in range(k):
    r = rows[i]
    print(f"  {r.name.decode():<20}{r.launches:>5} launches {r.total_ms * 1e3:>10.1f} us  {r.total_ms * 1e3 / max(r.launches, 1):>8.1f} us each  {100 * r.total_ms / tot:5.1f} %")
print(f"  sum (serialised) {tot * 1e3:.1f} us")
pipe.Destroy()
