"""Drop-in proof: the product HipForwardPipe compiled INSIDE the reference tree
(-DSAYURI_IN_TREE: the reference's own InputData / OutputResult / DNNWeights / DNNLoader, see
INTEGRATION.md and oracle/Makefile target libsayuri_ref_hip.so) gives the reference CPU pipe's
outputs on the same inputs.  Needs the prebuilt oracle/_ref library (built in the dev container,
shipped to the GPU box as a binary)."""
import ctypes
import os

import numpy as np
import pytest

from _golden import Golden
from _oracle import ORACLE_DIR
from sayuri_amd import _lib
from sayuri_amd import weights as W

pytestmark = pytest.mark.gpu
REF_HIP_SO = os.path.join(ORACLE_DIR, "_ref", "libsayuri_ref_hip.so")


@pytest.mark.skipif(not os.path.exists(REF_HIP_SO), reason="oracle/_ref/libsayuri_ref_hip.so not built")
def test_reference_tree_with_hip_pipe_matches_reference_cpu_pipe(tmp_weights_dir):
    _lib.hip()  # libsayuri_hip.so must be the product library already loaded in this process
    lib = ctypes.CDLL(REF_HIP_SO)
    lib.ref_last_error.restype = ctypes.c_char_p
    fp = ctypes.POINTER(ctypes.c_float)
    ip = ctypes.POINTER(ctypes.c_int)
    g = Golden("net_6b96", tmp_weights_dir)
    assert lib.ref_init(g.weights_path.encode(), 1) == 0, lib.ref_last_error()
    bsz = [19, 9, 13, 19, 19, 9]
    planes = W.synthetic_planes(len(bsz), bsz, seed=42)
    offs = [0, 1, 2, 3, 4, 0]
    # reference CPU pipe
    exp = []
    for p, bs, off in zip(planes, bsz, offs):
        out = np.zeros(2 * bs * bs + 9, np.float32)
        pc = np.ascontiguousarray(p, np.float32)
        assert lib.ref_forward(bs, ctypes.c_float(7.5), 0, off, pc.ctypes.data_as(fp), out.ctypes.data_as(fp)) == 0
        exp.append(out)
    for fp16, tol in ((0, 1e-4), (1, 4e-3)):
        assert lib.ref_hip_init(19, 8, fp16, 0) == 0, lib.ref_last_error()
        buf = np.zeros((len(bsz), 43 * 361), np.float32)
        for i, p in enumerate(planes):
            buf[i, :p.size] = p.ravel()
        out = np.zeros((len(bsz), 2 * 361 + 9), np.float32)
        b = np.asarray(bsz, np.int32)
        o = np.asarray(offs, np.int32)
        rc = lib.ref_hip_forward(len(bsz), b.ctypes.data_as(ip), o.ctypes.data_as(ip), buf.ctypes.data_as(fp),
                                 out.ctypes.data_as(fp))
        assert rc == 0, lib.ref_last_error()
        for i, bs in enumerate(bsz):
            s = bs * bs
            got = np.concatenate([out[i, :s], out[i, 361:361 + s], out[i, 722:]])
            gate = tol * (max(1.0, float(np.abs(exp[i]).max())) if fp16 else 1.0)  # fp16: relative to the output scale (test_gpu_net.py)
            assert np.abs(got - exp[i]).max() <= gate, (fp16, bs, float(np.abs(got - exp[i]).max()))
        lib.ref_hip_destroy()
