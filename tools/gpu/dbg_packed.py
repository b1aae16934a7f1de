"""Diagnose test_packed_planes_give_identical_outputs[tiny_all-True]: the test's sequence (raw fp32, raw packed, queue fp32 / packed / mixed)
many times in one process; prints every anomaly (which run, which samples, how far off)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from sayuri_amd import weights as W
from sayuri_amd.pipe import HipForwardPipe, hip_forward_packed_raw, hip_forward_raw
from sayuri_amd.engine import pack_planes
from _oracle import PortNet
import golden_specs, tempfile

name = sys.argv[1] if len(sys.argv) > 1 else "tiny_all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
fx = next(f for f in golden_specs.FIXTURES if f["name"] == name)
d = tempfile.mkdtemp()
path = os.path.join(d, name + ".bin")
W.write_weights(path, fx["spec"](), seed=fx["seed"])
bsz = [19, 9, 13, 19, 7, 19, 13, 19, 19, 9, 19]
planes = W.synthetic_planes(len(bsz), bsz, seed=31337)
for p in planes:
    p[37] = 1.0
    p[38] = 0.25
B = 19
grid = np.zeros((len(bsz), 43, B * B), np.float32)
for i, (p, bs) in enumerate(zip(planes, bsz)):
    grid[i].reshape(43, B, B)[:, :bs, :bs] = p.reshape(43, bs, bs)
records = np.stack([pack_planes(p, 37) for p in planes])
oracle = PortNet(path)
exp = [oracle.forward(p, bs) for p, bs in zip(planes, bsz)]
bad = 0
for rep in range(reps):
    pipe = HipForwardPipe(path, board_size=19, batch_size=16, fp16=True)
    try:
        ctx = pipe.ctx(0)
        a = hip_forward_raw(ctx, grid, bsz, B)
        b = hip_forward_packed_raw(ctx, records, 37, bsz, B)
        raw_same = all(np.array_equal(x, y) for x, y in zip(a, b))
        b0 = pipe.pump_times().get("batches", 0)
        runs = {}
        nb = {}
        for k, f in (("fp32", lambda: pipe.Forward(planes, bsz)), ("pack", lambda: pipe.ForwardPacked(planes, bsz)),
                     ("mix", lambda: pipe.ForwardPacked(planes, bsz, mixed=True))):
            runs[k] = f()
            b1 = pipe.pump_times().get("batches", 0)
            nb[k] = b1 - b0
            b0 = b1
        errs = {k: [float(np.abs(r[i] - exp[i]).max()) for i in range(len(bsz))] for k, r in runs.items()}
        worst = max(max(e) for e in errs.values())
        if worst > 0.05 or not raw_same:
            bad += 1
            print("rep", rep, "raw_same", raw_same, "batches per run", nb)
            for k, e in errs.items():
                print("   %-5s %s" % (k, " ".join("%.1e" % x for x in e)))
    finally:
        pipe.Destroy()
print("anomalies: %d / %d" % (bad, reps), os.environ.get("TAG", ""))
