"""40b x 384 network, batch 256 of 19x19 boards: the two channel tiles that divide 384 (A/B via SAYURI_BOARD_KOT)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sayuri_amd import _lib, weights as W
from sayuri_amd.pipe import HipForwardPipe
lib = _lib.hip()
path = "/tmp/kot384_w.bin"
if not os.path.exists(path):
    W.write_weights(path, W.spec_40b384(), seed=3)
n = 256
pipe = HipForwardPipe(path, board_size=19, batch_size=n, fp16=True)
ctx = pipe.ctx(0)
grid = np.ascontiguousarray(np.stack(W.synthetic_planes(n, 19, seed=1)), np.float32)
bsz = np.full(n, 19, np.int32)
assert lib.sayuri_hip_upload(ctx, n, grid.ctypes.data_as(_lib.c_float_p), bsz.ctypes.data_as(_lib.c_int_p)) == 0
ms = ctypes.c_float(0)
lib.sayuri_hip_mark_kernel(ctx, b"")
lib.sayuri_hip_time_runs(ctx, 2, ctypes.byref(ms))
lib.sayuri_hip_mark_kernel(ctx, b"conv3x3_tower/5")
lib.sayuri_hip_time_runs(ctx, 6, ctypes.byref(ms))
st = _lib.KernelStat()
lib.sayuri_hip_timed_stat(ctx, ctypes.byref(st))
print(os.environ.get("SAYURI_BOARD_KOT", "default"), "ms/step", round(ms.value / 6, 3), "evals/s", round(n * 6 / ms.value * 1e3), "tower us", round(st.total_ms / st.launches * 1e3, 1),
      "TF", round(st.flops / st.launches / (st.total_ms / st.launches * 1e-3) / 1e12, 1))
pipe.Destroy()
