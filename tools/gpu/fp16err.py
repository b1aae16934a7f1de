"""fp16 engine error vs the reference goldens, per fixture: max abs error, output scale, SelfCheck L2."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _golden import Golden
from golden_specs import FIXTURES
from sayuri_amd.pipe import HipForwardPipe
from test_gpu_net import self_check_l2
d = tempfile.mkdtemp()
for fx in FIXTURES:
    g = Golden(fx["name"], d)
    cases = [c for c in g.cases if c["winograd"] == 1]
    pipe = HipForwardPipe(g.weights_path, board_size=19, batch_size=64, fp16=True)
    outs = pipe.BatchForward([g.planes(c) for c in cases], [c["board_size"] for c in cases], offsets=[c["offset"] for c in cases])
    pipe.Destroy()
    errs = [float(np.abs(o - g.expected(c)).max()) for o, c in zip(outs, cases)]
    scale = max(float(np.abs(g.expected(c)).max()) for c in cases)
    l2 = max(self_check_l2(o, g.expected(c), c["board_size"]) for o, c in zip(outs, cases))
    print(f"{fx['name']:<22} cases {len(cases):>3}  max abs err {max(errs):.5f}  output scale {scale:8.3f}  rel {max(errs)/scale:.2e}  selfcheck L2 {l2:.2e}")
