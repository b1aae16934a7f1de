#!/usr/bin/env python3
"""Chains in ONE set of activation buffers (SAYURI_CHAINS_OWN_BUFS=0) on a SHORT 384-channel network, so that every layer's
output is still in a buffer after the forward: which buffer / rows / bytes differ from the one-chain forward when a run goes
wrong, and what do the wrong bytes look like (another layer's rows? zeros? 64-byte halves?).  Debugging aid."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sayuri_amd import _lib, weights as W
from sayuri_amd.pipe import HipForwardPipe, hip_forward_raw
POISON = False  # (needed a fill tap that is not in the library any more)
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
spec = W.NetSpec.residual(blocks, 384, 48)
for b in spec.blocks:
    b.se = False
wpath = f"/tmp/sayuri_dbg_{blocks}b384.bin"
W.write_weights(wpath, spec, seed=5)
rng = np.random.default_rng(56)
n, B = 256, 19
bsz = [int(b) for b in rng.choice([9, 13, 19], size=n)]
planes = W.synthetic_planes(n, bsz, seed=5856)
grid = np.zeros((n, 43, B * B), np.float32)
for i, (p, bs) in enumerate(zip(planes, bsz)):
    grid[i].reshape(43, B, B)[:, :bs, :bs] = p.reshape(43, bs, bs)
order = np.argsort(-np.array(bsz), kind="stable")
dev_bsz = [bsz[i] for i in order]
lib = _lib.hip()
lib.sayuri_hip_debug_read_activations.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
if hasattr(lib, 'sayuri_hip_debug_fill_activations'): lib.sayuri_hip_debug_fill_activations.argtypes = [ctypes.c_void_p, ctypes.c_int]
CS = 384
ROW = CS * 2
SLOT = 361 * ROW
def bufs(ctx):
    out = []
    for b in range(6):
        a = np.zeros(n * SLOT, np.uint8)
        assert lib.sayuri_hip_debug_read_activations(ctx, b, a.ctypes.data, a.size) == 0
        out.append(a.reshape(n, 361, ROW))
    return out
def run(env, reps):
    for k in ("SAYURI_CHAINS", "SAYURI_CHAINS_OWN_BUFS", "SAYURI_CHAINS_SERIAL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    pipe = HipForwardPipe(wpath, board_size=B, batch_size=256, fp16=True)
    res = []
    global PREV
    for r in range(reps):
        if POISON:
            lib.sayuri_hip_debug_fill_activations(pipe.ctx(0), 0xFF)   # fp16 NaN everywhere: a row read before it is written shows
        o = hip_forward_raw(pipe.ctx(0), grid, bsz, B)
        if POISON and not all(np.isfinite(x).all() for x in o):
            nans = [i for i in range(n) if not all(np.isfinite(x[i]).all() for x in o)]
            dev = sorted(int(np.nonzero(order == i)[0][0]) for i in nans)
            print(f"run {r}: NaN outputs for {len(nans)} samples; device positions {dev[:8]} .. {dev[-3:]}")
        res.append((o, bufs(pipe.ctx(0)) if (r == 0 or env.get("SAYURI_CHAINS") != "1") else None))
        PREV = res[-2][1] if len(res) >= 2 else None
        if env.get("SAYURI_CHAINS") != "1" and r > 0:
            good = all(np.array_equal(a, b) for a, b in zip(REF[0][0], o))
            if not good:
                print(f"run {r}: outputs differ; chains {lib.sayuri_hip_last_chains(pipe.ctx(0))}")
                cur = res[-1][1]
                for b in range(6):
                    d = (cur[b] != REFB[b])
                    if not d.any():
                        continue
                    samples = np.nonzero(d.any(axis=(1, 2)))[0]
                    print(f"  buffer {b}: {len(samples)} device samples differ: {samples[:16].tolist()}{'...' if len(samples) > 16 else ''}")
                    s0 = int(samples[0]); bs = dev_bsz[s0]
                    rows = np.nonzero(d[s0].any(axis=1))[0]
                    print(f"    sample {s0} (board {bs}): {len(rows)} of {bs * bs} pixel rows differ: {rows[:20].tolist()}")
                    r0 = int(rows[0])
                    bytes_ = np.nonzero(d[s0, r0])[0]
                    print(f"    row {r0}: differing byte range {int(bytes_[0])}..{int(bytes_[-1])} ({len(bytes_)} bytes); channel-tile thirds hit: {sorted(set(int(x) // 256 for x in bytes_))}; 64-byte pieces hit: {sorted(set(int(x) // 64 for x in bytes_))[:12]}")
                    got = cur[b][s0, r0].view(np.float16).astype(np.float32); exp = REFB[b][s0, r0].view(np.float16).astype(np.float32)
                    print(f"    max |got - exp| {np.abs(got - exp).max():.4g} on values of scale {np.abs(exp).max():.3g}; got zeros: {(got == 0).mean():.2f}")
                    # is the wrong sample another SAMPLE's data of the reference run (same buffer)?  compare the first 8 rows
                    for s2 in range(n):
                        if s2 != s0 and dev_bsz[s2] == bs and np.array_equal(cur[b][s0, :8], REFB[b][s2, :8]):
                            print(f"    -> sample {s0}'s rows in buffer {b} are sample {s2}'s rows of the reference run")
                            break
                    # is it the reference data of the same sample in the PREVIOUS forward's state (i.e. not rewritten this forward)?
                    if PREV is not None and np.array_equal(cur[b][s0, :8], PREV[b][s0, :8]):
                        print(f"    -> sample {s0}'s rows in buffer {b} are what the buffer held BEFORE this forward")
                    # do the wrong bytes equal the same place in ANOTHER buffer of the reference run (= another layer's output)?
                    for ob in range(6):
                        if ob != b and np.array_equal(cur[b][s0, r0][bytes_], REFB[ob][s0, r0][bytes_]):
                            print(f"    -> the wrong bytes are buffer {ob}'s bytes of the same row in the reference run")
                break
        if len(res) > 2:
            res.pop(0)
    pipe.Destroy()
    return res
PREV = None
REF = run({"SAYURI_CHAINS": "1"}, 1)
REFB = REF[0][1]
print("reference buffers in use:", [int((b != 0).any()) for b in REFB])
if "own" in sys.argv:   # every chain but the first in buffers of its own: outputs only (ticket 0's buffers hold chain 0's rows)
    os.environ.pop("SAYURI_CHAINS_OWN_BUFS", None); os.environ["SAYURI_CHAINS"] = "3"
    pipe = HipForwardPipe(wpath, board_size=B, batch_size=256, fp16=True)
    bad = 0
    for r in range(reps):
        o = hip_forward_raw(pipe.ctx(0), grid, bsz, B)
        bad += not all(np.array_equal(a, b) for a, b in zip(REF[0][0], o))
    print(f"own buffers: {bad} of {reps} forwards differ from the one-chain outputs ({lib.sayuri_hip_last_chains(pipe.ctx(0))} chains)")
    pipe.Destroy()
else:
    run({"SAYURI_CHAINS": "3", "SAYURI_CHAINS_OWN_BUFS": "0"}, reps)
print("done")
