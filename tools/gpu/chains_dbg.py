#!/usr/bin/env python3
"""Which samples differ between the one-chain forward and a G-chain forward (debugging aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sayuri_amd import _lib, weights as W
from sayuri_amd.pipe import HipForwardPipe, hip_forward_raw
wpath = "/tmp/sayuri_c5_40b384.bin"
if not os.path.exists(wpath):
    W.write_weights(wpath, W.spec_40b384(), seed=23)
rng = np.random.default_rng(56)
n, B = 256, 19
bsz = [int(b) for b in rng.choice([9, 13, 19], size=n)]
planes = W.synthetic_planes(n, bsz, seed=5600 + n)
grid = np.zeros((n, 43, B * B), np.float32)
for i, (p, bs) in enumerate(zip(planes, bsz)):
    grid[i].reshape(43, B, B)[:, :bs, :bs] = p.reshape(43, bs, bs)
order = np.argsort(-np.array(bsz), kind="stable")   # device order: largest first
rank = np.empty(n, int); rank[order] = np.arange(n)
def run(mode):
    os.environ["SAYURI_CHAINS"] = mode
    pipe = HipForwardPipe(wpath, board_size=B, batch_size=256, fp16=True)
    try:
        outs = [hip_forward_raw(pipe.ctx(0), grid, bsz, B) for _ in range(3)]
        return outs, _lib.hip().sayuri_hip_last_chains(pipe.ctx(0))
    finally:
        pipe.Destroy()
ref, _ = run("1")
print("one chain, run to run:", [bool(all(np.array_equal(a, b) for a, b in zip(ref[0], r))) for r in ref[1:]])
for mode in sys.argv[1:] or ["2", "3", "4", "2"]:
    outs, g = run(mode)
    for k, o in enumerate(outs):
        bad = [i for i in range(n) if not all(np.array_equal(a[i], b[i]) for a, b in zip(ref[0], o))]
        dev = sorted(int(rank[i]) for i in bad)
        print(f"SAYURI_CHAINS={mode} ({g} chains) run {k}: {len(bad)} samples differ; device positions {dev[:12]}{'...' if len(dev) > 12 else ''}"
              + (f" .. {dev[-3:]}; sizes {sorted(set(bsz[i] for i in bad))}; max diff {max(float(np.abs(ref[0][0][i] - o[0][i]).max()) for i in bad):.3g}" if bad else ""))
