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


REF_HIP_SC_SO = os.path.join(ORACLE_DIR, "_ref", "libsayuri_ref_hip_sc.so")


def _callers(path):
    _lib.hip()
    lib = ctypes.CDLL(path)
    lib.ref_hip_callers_error.restype = ctypes.c_char_p
    lib.ref_hip_net_new.restype = ctypes.c_void_p
    lib.ref_hip_net_new.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 5
    lib.ref_hip_net_free.argtypes = [ctypes.c_void_p]
    dp = ctypes.POINTER(ctypes.c_double)
    lib.ref_hip_netbench.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, dp]
    lib.ref_hip_search.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, dp]
    return lib


@pytest.mark.skipif(not os.path.exists(REF_HIP_SO), reason="oracle/_ref/libsayuri_ref_hip.so not built")
def test_reference_netbench_on_the_hip_pipe(tmp_weights_dir):
    """The reference's own Network facade (encoder, symmetry, Network::GetOutput, network.cc:237-291) called from 512
    threads of the reference's ThreadPool -- the loop of its GTP `netbench` (gtp.cc:1516-1557) -- over HipForwardPipe on
    the 20b x 256 network: the plugin interface's threading contract from the reference side, at full speed."""
    lib = _callers(REF_HIP_SO)
    g = Golden("net_20b256", tmp_weights_dir)
    net = lib.ref_hip_net_new(g.weights_path.encode(), 19, 256, 1, 0, 5)
    assert net, lib.ref_hip_callers_error()
    try:
        out = np.zeros(5, np.float64)
        dp = out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        assert lib.ref_hip_netbench(net, 19, 512, ctypes.c_float(2.0), dp) == 0, lib.ref_hip_callers_error()  # warm-up
        assert lib.ref_hip_netbench(net, 19, 512, ctypes.c_float(8.0), dp) == 0, lib.ref_hip_callers_error()
        evals, secs, rate, batches, mean_batch = out
        print(f"reference netbench over HipForwardPipe: {rate:.0f} evals/s, mean batch {mean_batch:.1f}, {int(batches)} batches")
        # the reference's encoder runs on the caller threads (260-490 us per position): what the box's host
        # cores can encode bounds the rate, so the floor here is the queue's health (full batches), not the GPU's ceiling
        assert mean_batch > 200, out
        assert rate > 30000, out
    finally:
        lib.ref_hip_net_free(net)


@pytest.mark.skipif(not os.path.exists(REF_HIP_SC_SO), reason="oracle/_ref/libsayuri_ref_hip_sc.so not built")
@pytest.mark.parametrize("fp16", [0, 1], ids=["fp32", "fp16"])
def test_reference_search_with_8_threads_self_checked(tmp_weights_dir, fp16):
    """The reference's Search::Computation with threads=8 (search.cc:252-436: eight playouts in flight in one tree, virtual
    loss, every leaf a blocking Forward call from a pool thread) over HipForwardPipe, in the build with the reference's own
    -DSELF_CHECK: every evaluation is repeated on the reference CPU pipe and compared (L2 <= 0.2, network.cc:333-359) -- a
    mismatch throws and fails the search."""
    lib = _callers(REF_HIP_SC_SO)
    g = Golden("net_6b96", tmp_weights_dir)
    net = lib.ref_hip_net_new(g.weights_path.encode(), 19, 8, fp16, 0, 5)
    assert net, lib.ref_hip_callers_error()
    try:
        moves = np.asarray([3 * 19 + 3, 15 * 19 + 15, 3 * 19 + 15, 15 * 19 + 3, 9 * 19 + 9], np.int32)
        out = np.zeros(5, np.float64)
        rc = lib.ref_hip_search(net, 19, ctypes.c_float(7.5), moves.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(moves), 8, 600,
                                out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        assert rc == 0, lib.ref_hip_callers_error()
        best, visits, playouts, secs, queries = out
        print(f"reference search, 8 threads: best {int(best)}, {int(visits)} visits, {int(playouts)} playouts, {int(queries)} NN queries in {secs:.2f} s")
        assert 0 <= best <= 361 and playouts >= 600 and queries > 300
    finally:
        lib.ref_hip_net_free(net)
