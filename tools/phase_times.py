#!/usr/bin/env python3
"""Time the three phases of sayuri_hip_forward (upload / run / download) on an idle host."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sayuri_amd import _lib, weights as W
from sayuri_amd.pipe import HipForwardPipe
wpath = f"/tmp/sayuri_bench_20b256_seed22_{os.getuid()}.bin"
if not os.path.exists(wpath):
    W.write_weights(wpath, W.spec_20b256(), seed=22)
n = 256
pipe = HipForwardPipe(wpath, board_size=19, batch_size=n, fp16=True, device=0)
ctx = pipe.ctx(0); lib = _lib.hip()
lib.sayuri_hip_host_alloc.restype = ctypes.c_void_p
lib.sayuri_hip_host_alloc.argtypes = [ctypes.c_size_t]
def pinned(count):
    p = lib.sayuri_hip_host_alloc(count * 4)
    return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_float)), shape=(count,))
planes = pinned(n * 43 * 361); planes[:] = np.random.default_rng(0).random(n * 43 * 361, dtype=np.float32)
prob, pas, misc, own = pinned(n * 5 * 361), pinned(n * 5), pinned(n * 15), pinned(n * 361)
bsz = np.full(n, 19, np.int32)
fp = lambda a: a.ctypes.data_as(_lib.c_float_p)
for name, pl in (("pinned", planes), ("pageable", planes.copy())):
    tu = tr = td = 0.0
    for it in range(12):
        t0 = time.perf_counter(); lib.sayuri_hip_upload(ctx, n, fp(pl), bsz.ctypes.data_as(_lib.c_int_p))
        t1 = time.perf_counter(); lib.sayuri_hip_run(ctx); lib.sayuri_hip_sync(ctx)
        t2 = time.perf_counter(); lib.sayuri_hip_download(ctx, fp(prob), fp(pas), fp(misc), fp(own))
        t3 = time.perf_counter()
        if it >= 2: tu += t1 - t0; tr += t2 - t1; td += t3 - t2
    print(f"{name}: upload {tu*100:.3f} ms  run+sync {tr*100:.3f} ms  download {td*100:.3f} ms")
pipe.Destroy()
