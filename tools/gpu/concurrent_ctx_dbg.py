#!/usr/bin/env python3
"""Do two INDEPENDENT contexts (own buffers, own streams) running their forwards at the same time still give the bits they
give alone?  (debugging aid for the chained forward: per-layer launches of one queue beside per-layer launches of another)"""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sayuri_amd import _lib, weights as W
from sayuri_amd.pipe import HipForwardPipe, hip_forward_raw
net = sys.argv[1] if len(sys.argv) > 1 else "40b384"
uniform = "--uniform" in sys.argv
spec = {"40b384": W.spec_40b384, "20b256": W.spec_20b256}[net]()
wpath = f"/tmp/sayuri_c5_{net}.bin"
if not os.path.exists(wpath):
    W.write_weights(wpath, spec, seed=23)
os.environ["SAYURI_CHAINS"] = "1"
rng = np.random.default_rng(56)
n, B = 256, 19
bsz = [int(b) for b in rng.choice([9, 13, 19], size=n)]
if uniform:
    bsz = [19] * n
if "--partial" in sys.argv:   # a partial batch of full boards: 250 workgroups on 256 CUs
    n = 250
    bsz = bsz[:n]
planes = W.synthetic_planes(n, bsz, seed=5600 + n)
grid = np.zeros((n, 43, B * B), np.float32)
for i, (p, bs) in enumerate(zip(planes, bsz)):
    grid[i].reshape(43, B, B)[:, :bs, :bs] = p.reshape(43, bs, bs)
pipes = [HipForwardPipe(wpath, board_size=B, batch_size=256, fp16=True) for _ in range(2)]
ref = hip_forward_raw(pipes[0].ctx(0), grid, bsz, B)
ref2 = hip_forward_raw(pipes[1].ctx(0), grid, bsz, B)
print("alone: ctx0 == ctx1:", all(np.array_equal(a, b) for a, b in zip(ref, ref2)))
res = {}
def work(k, reps):
    bad = 0
    for _ in range(reps):
        o = hip_forward_raw(pipes[k].ctx(0), grid, bsz, B)
        bad += sum(1 for i in range(n) if not all(np.array_equal(a[i], b[i]) for a, b in zip(ref, o)))
    res[k] = bad
ths = [threading.Thread(target=work, args=(k, 6)) for k in range(2)]
for t in ths: t.start()
for t in ths: t.join()
print(f"{net}{' uniform 19x19' if uniform else ' mixed'}: two contexts at once, 6 forwards each: samples that differ from the solo result: {res}")
# the production path: ONE context, two tickets in flight through submit / wait (fp32 planes from pinned buffers)
import ctypes
lib = _lib.hip()
FP = ctypes.POINTER(ctypes.c_float)
lib.sayuri_hip_host_alloc.restype = ctypes.c_void_p
lib.sayuri_hip_host_alloc.argtypes = [ctypes.c_size_t]
lib.sayuri_hip_submit.argtypes = [ctypes.c_void_p, ctypes.c_int, FP, ctypes.POINTER(ctypes.c_int), FP, FP, FP, FP, ctypes.POINTER(ctypes.c_int)]
lib.sayuri_hip_wait.argtypes = [ctypes.c_void_p, ctypes.c_int]
ctx = pipes[0].ctx(0)
sizes = (grid.size, n * 5 * B * B, n * 5, n * 15, n * B * B)
bufs = []
for _ in range(2):
    ptrs = [lib.sayuri_hip_host_alloc(k * 4) for k in sizes]
    np.ctypeslib.as_array(ctypes.cast(ptrs[0], FP), (grid.size,))[:] = grid.ravel()
    bufs.append(ptrs)
hb = lib.sayuri_hip_host_alloc(n * 4)
np.ctypeslib.as_array(ctypes.cast(hb, ctypes.POINTER(ctypes.c_int32)), (n,))[:] = np.asarray(bsz, np.int32)
bp = ctypes.cast(hb, ctypes.POINTER(ctypes.c_int))
tick = [ctypes.c_int(-1), ctypes.c_int(-1)]
def submit(i):
    pl, pr, pa, mi, ow = bufs[i]
    assert lib.sayuri_hip_submit(ctx, n, ctypes.cast(pl, FP), bp, ctypes.cast(pr, FP), ctypes.cast(pa, FP), ctypes.cast(mi, FP), ctypes.cast(ow, FP), ctypes.byref(tick[i])) == 0
def check(i):
    assert lib.sayuri_hip_wait(ctx, tick[i].value) == 0
    pl, pr, pa, mi, ow = bufs[i]
    got = (np.ctypeslib.as_array(ctypes.cast(pr, FP), (n, 5, B * B)), np.ctypeslib.as_array(ctypes.cast(pa, FP), (n, 5)),
           np.ctypeslib.as_array(ctypes.cast(mi, FP), (n, 15)), np.ctypeslib.as_array(ctypes.cast(ow, FP), (n, B * B)))
    return sum(1 for s_ in range(n) if not all(np.array_equal(a[s_], b[s_]) for a, b in zip(ref, got)))
bad = []
submit(0); submit(1)
nb = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--batches=")), 12)
for k in range(nb - 2):
    bad.append(check(k & 1)); submit(k & 1)
bad.append(check(0)); bad.append(check(1))
print(f"{net}: one context, two tickets in flight (submit / wait), {len(bad)} batches: {sum(1 for b in bad if b)} batches differ from the solo result; samples per batch: {bad if len(bad) <= 16 else sorted(set(bad))}")
for p in pipes: p.Destroy()
