#!/usr/bin/env python3
"""Reference-style netbench through the whole pipe (queue + staging + H2D/D2H), 20b256 19x19."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sayuri_amd import weights as W
from sayuri_amd.pipe import HipForwardPipe
wpath = f"/tmp/sayuri_bench_20b256_seed22_{os.getuid()}.bin"
if not os.path.exists(wpath):
    W.write_weights(wpath, W.spec_20b256(), seed=22)
for batch in (int(a) for a in (sys.argv[1:] or ["256"])):
    pipe = HipForwardPipe(wpath, board_size=19, batch_size=batch, fp16=True, device=0, waittime_ms=2)
    eps, tot = pipe.netbench(threads=2 * batch, seconds=4.0)
    print(f"netbench batch={batch} threads={2*batch}: {tot} evals | {eps:.1f} evals/s")
    t = pipe.pump_times()
    nb = max(t["batches"], 1)
    print(f"   batches={t['batches']} avg_batch={t['evals']/nb:.1f} per-batch us: forward={t['forward_us']/nb:.0f} "
          f"fill={t['fill_us']/nb:.0f} wait_batch={t['wait_batch_us']/nb:.0f} wait_copies={t['wait_copies_us']/nb:.0f}")
    pipe.Destroy()
