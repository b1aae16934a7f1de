"""Layer-level parity of the HIP kernels (through the C-ABI tap sayuri_hip_test_conv) against a
float64 numpy restatement of the direct convolution the CPU oracle computes
(reference src/neural/blas/convolution.h:41-125, convolution.cc:27-62, biases.cc:14-77).

fp32 mode: fp32 MFMA is an exact fmaf chain -> abs tolerance 2e-5 on O(1) outputs.
fp16 mode: inputs/weights rounded to fp16, fp32 accumulate, fp16 store -> tolerance
4e-3 * max|y| (half has 11 significant bits; K <= 2304 products of O(1)*O(0.03))."""
import ctypes

import numpy as np
import pytest

from sayuri_amd import _lib

pytestmark = pytest.mark.gpu


def _fp(a):
    return a.ctypes.data_as(_lib.c_float_p)


def act_np(x, act):
    """The eight activations of the reference (src/neural/activation.h:36-81), float64."""
    if act == 0:
        return x
    if act == 1:
        return np.maximum(x, 0)
    if act == 2:  # ELU
        return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))
    if act == 3:  # SELU
        return np.where(x > 0, 1.05070098 * x, 1.05070098 * 1.67326324 * np.expm1(np.minimum(x, 0)))
    if act == 4:  # GELU, tanh form
        return 0.5 * x * (1 + np.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3)))
    if act == 5:
        return x * np.tanh(np.log1p(np.exp(x)))
    if act == 6:
        return x / (1 + np.exp(-x))
    if act == 7:  # HardSwish
        return np.where(x >= 3, x, np.where(x <= -3, 0.0, x * (x + 3) / 6))
    raise ValueError(act)


def conv_ref(xs, bsz, w, bias, res, k, depthwise, act, post):
    """xs: list of [C][bs*bs] arrays; returns list of [K][bs*bs] float64."""
    outs = []
    pad = k // 2
    for i, (x, bs) in enumerate(zip(xs, bsz)):
        C = x.shape[0]
        img = np.zeros((C, bs + 2 * pad, bs + 2 * pad))
        img[:, pad:pad + bs, pad:pad + bs] = x.reshape(C, bs, bs)
        K = w.shape[0]
        y = np.zeros((K, bs, bs))
        for kr in range(k):
            for kc in range(k):
                patch = img[:, kr:kr + bs, kc:kc + bs]
                if depthwise:
                    y += patch * w[:, 0, kr, kc][:, None, None]
                else:
                    y += np.einsum("kc,cyx->kyx", w[:, :, kr, kc].astype(np.float64), patch)
        y = y.reshape(K, bs * bs)
        if bias is not None:
            y = y + bias[:, None]
        if post:
            y = act_np(y, act)
            if res is not None:
                y = y + res[i]
        else:
            if res is not None:
                y = y + res[i]
            y = act_np(y, act)
        outs.append(y)
    return outs


def run_case(fp16, bsz, cin, cout, k, depthwise=False, act=5, with_res=True, post=False, seed=0, max_board=19, kind=None):
    rng = np.random.default_rng(seed)
    n = len(bsz)
    xc = cout if depthwise else cin
    xs = [rng.standard_normal((xc, b * b)).astype(np.float32) for b in bsz]
    wshape = (cout, 1, k, k) if depthwise else (cout, cin, k, k)
    fan = k * k * (1 if depthwise else cin)
    w = (rng.standard_normal(wshape) / np.sqrt(fan)).astype(np.float32)
    bias = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    res = [rng.standard_normal((cout, b * b)).astype(np.float32) for b in bsz] if with_res else None
    if fp16:  # the kernel sees fp16-rounded operands; give the reference the same
        xs_r = [x.astype(np.float16).astype(np.float64) for x in xs]
        w_r = w.astype(np.float16).astype(np.float64) if not depthwise else w.astype(np.float64)
        res_r = [r.astype(np.float16).astype(np.float64) for r in res] if res else None
    else:
        xs_r, w_r = [x.astype(np.float64) for x in xs], w.astype(np.float64)
        res_r = [r.astype(np.float64) for r in res] if res else None
    ref = conv_ref(xs_r, bsz, w_r, bias.astype(np.float64), res_r, k, depthwise, act, post)
    xcat = np.concatenate([x.ravel() for x in xs])
    rcat = np.concatenate([r.ravel() for r in res]) if res else None
    y = np.zeros(sum(cout * b * b for b in bsz), np.float32)
    bs_arr = np.asarray(bsz, np.int32)
    rc = _lib.hip().sayuri_hip_test_conv(0, int(fp16), n, bs_arr.ctypes.data_as(_lib.c_int_p), max_board, cin, cout, k,
                                         int(depthwise), act, int(post), _fp(xcat), _fp(w.ravel()), _fp(bias),
                                         _fp(rcat) if res else None, _fp(y))
    assert rc == 0, _lib.hip().sayuri_hip_last_error().decode()
    off = 0
    worst = 0.0
    scale = max(float(np.abs(r).max()) for r in ref)
    for i, b in enumerate(bsz):
        got = y[off:off + cout * b * b].reshape(cout, b * b)
        off += cout * b * b
        assert np.isfinite(got).all()
        worst = max(worst, float(np.abs(got - ref[i]).max()))
    tol = 4e-3 * scale if fp16 else 2e-5 * max(scale, 1.0)
    assert worst <= tol, (worst, tol, scale)
    if kind is not None:
        assert _lib.hip().sayuri_hip_test_last_conv_kind() == kind, "the layer ran on another kernel family than the test is about"
    return worst


CASES = [
    # bsz, cin, cout, k
    ([19], 32, 32, 3),
    ([19, 19, 19], 64, 64, 3),
    ([9, 13, 19, 7, 19], 32, 64, 3),       # mixed boards in one batch, tiles crossing samples
    ([19] * 4, 43, 96, 3),                   # input conv shape of the 6b96 net (cin padded to 64)
    ([19] * 3, 96, 96, 3),
    ([19] * 2, 256, 256, 3),                 # the tower conv of the 20b256 net
    ([13] * 5, 128, 192, 3),
    ([19] * 2, 256, 32, 1),                  # head conv
    ([9, 19], 48, 72, 1),                    # mixer ffn-like 1x1 with odd channel counts
    ([19] * 2, 384, 384, 3),                 # 40b384 tower conv (two ko tiles)
    ([2, 3, 5, 19], 32, 32, 3),              # tiny boards
]


@pytest.mark.parametrize("fp16", [False, True], ids=["fp32", "fp16"])
@pytest.mark.parametrize("case", CASES, ids=[f"{c[1]}x{c[2]}k{c[3]}n{len(c[0])}b{min(c[0])}" for c in CASES])
def test_conv_mfma(case, fp16):
    bsz, cin, cout, k = case
    run_case(fp16, bsz, cin, cout, k, act=5, with_res=True, seed=cin + cout)


@pytest.mark.parametrize("fp16", [False, True], ids=["fp32", "fp16"])
@pytest.mark.parametrize("act", range(8))
def test_conv_epilogue_variants(act, fp16):
    run_case(fp16, [19, 9], 32, 32, 3, act=act, with_res=False, seed=act)
    run_case(fp16, [19, 9], 32, 32, 3, act=act, with_res=True, seed=act + 10)


KIND_BOARD = 2
BOARD_CASES = [
    # bsz, cin, cout: fp16 3x3 layers the one-workgroup-per-board kernel (conv_board.h) takes
    ([19] * 3, 256, 256),                         # the tower conv of the 20b256 net, one board per tile (12 + 11 column tiles)
    ([19] * 4, 43, 256),                          # its input conv (cin padded to 64: two chunks)
    ([19] * 2, 384, 384),                         # 40b384: two 192-channel tiles per board (odd row-tile count per wave)
    ([13] * 5, 128, 192),                         # two boards per tile + a half-empty last tile
    ([9] * 9, 64, 128),                           # four boards per tile, 128-channel tile (two row tiles per wave)
    ([19, 19, 13, 13, 9, 9, 9, 9, 19, 13], 256, 256),   # mixed sizes: one size per tile
    ([7] * 13, 32, 128),                          # seven boards per tile
    ([19], 256, 256),                             # a batch of one
]


@pytest.mark.parametrize("case", BOARD_CASES, ids=[f"{c[1]}x{c[2]}n{len(c[0])}b{min(c[0])}" for c in BOARD_CASES])
def test_conv_board(case):
    bsz, cin, cout = case
    run_case(True, bsz, cin, cout, 3, act=5, with_res=True, seed=cin + cout, kind=KIND_BOARD)
    run_case(True, bsz, cin, cout, 3, act=0, with_res=False, seed=cin + cout + 1, kind=KIND_BOARD)


@pytest.mark.parametrize("act", range(8))
def test_conv_board_activations(act):
    run_case(True, [19, 19], 64, 128, 3, act=act, with_res=True, seed=40 + act, kind=KIND_BOARD)
    run_case(True, [13, 13, 13], 64, 192, 3, act=act, with_res=False, seed=50 + act, kind=KIND_BOARD)


def test_conv_board_batch256():
    """Full bench geometry (256 x 19x19 = 256 board tiles = one workgroup per CU): a subset of samples against the float64
    reference, and sample independence -- permuting the batch permutes the outputs bit for bit."""
    rng = np.random.default_rng(17)
    n, cin, cout = 256, 64, 256
    x = rng.standard_normal((n, cin, 361)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    bias = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    bs_arr = np.full(n, 19, np.int32)
    lib = _lib.hip()

    def run(xx):
        y = np.zeros((n, cout, 361), np.float32)
        rc = lib.sayuri_hip_test_conv(0, 1, n, bs_arr.ctypes.data_as(_lib.c_int_p), 19, cin, cout, 3, 0, 5, 0,
                                      _fp(np.ascontiguousarray(xx).ravel()), _fp(w.ravel()), _fp(bias), None, _fp(y.ravel()))
        assert rc == 0, lib.sayuri_hip_last_error().decode()
        assert lib.sayuri_hip_test_last_conv_kind() == KIND_BOARD
        return y

    y = run(x)
    w16 = w.astype(np.float16).astype(np.float64)
    for i in (0, 1, 99, 100, 177, 255):
        ref = conv_ref([x[i].astype(np.float16).astype(np.float64)], [19], w16, bias.astype(np.float64), None, 3, False, 5,
                       False)[0]
        assert np.abs(y[i] - ref).max() <= 4e-3 * np.abs(ref).max()
    perm = rng.permutation(n)
    np.testing.assert_array_equal(run(x[perm]), y[perm])


@pytest.mark.parametrize("fp16", [False, True], ids=["fp32", "fp16"])
@pytest.mark.parametrize("k", [3, 5, 7])
def test_depthwise(k, fp16):
    run_case(fp16, [19, 9, 13], 1, 48, k, depthwise=True, act=5, with_res=True, post=True, seed=k)
    run_case(fp16, [19, 7], 1, 32, k, depthwise=True, act=1, with_res=False, seed=k + 1)


def test_conv_batch256_tile_seams():
    """Full bench geometry (256 x 19x19): every pixel tile seam / sample crossing is exercised;
    checked through linearity in the input (conv(a*x) = a*conv(x) with identity act, no bias)
    and against the float64 reference on a subset of samples."""
    rng = np.random.default_rng(5)
    n, cin, cout = 256, 32, 32
    x = rng.standard_normal((n, cin, 361)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    bs_arr = np.full(n, 19, np.int32)
    y = np.zeros((n, cout, 361), np.float32)
    lib = _lib.hip()
    rc = lib.sayuri_hip_test_conv(0, 0, n, bs_arr.ctypes.data_as(_lib.c_int_p), 19, cin, cout, 3, 0, 0, 0,
                                  _fp(x.ravel()), _fp(w.ravel()), None, None, _fp(y.ravel()))
    assert rc == 0, lib.sayuri_hip_last_error().decode()
    for i in (0, 1, 100, 177, 255):
        ref = conv_ref([x[i].astype(np.float64)], [19], w.astype(np.float64), None, None, 3, False, 0, False)[0]
        assert np.abs(y[i] - ref).max() < 2e-5
    # samples are independent: permuting the batch permutes the outputs
    perm = rng.permutation(n)
    y2 = np.zeros_like(y)
    rc = lib.sayuri_hip_test_conv(0, 0, n, bs_arr.ctypes.data_as(_lib.c_int_p), 19, cin, cout, 3, 0, 0, 0,
                                  _fp(np.ascontiguousarray(x[perm]).ravel()), _fp(w.ravel()), None, None, _fp(y2.ravel()))
    assert rc == 0
    np.testing.assert_array_equal(y2, y[perm])


@pytest.mark.parametrize("env", [("SAYURI_CONV", "glds"), ("SAYURI_CONV", "v0")], ids=lambda e: f"{e[0]}={e[1]}")
def test_conv_kernel_variants(env):
    """The A/B switch of the fp16 3x3 kernels (glds: tiles across samples instead of one workgroup per board; v0: the
    generic register-staged kernel) is read once per process, so each runs the fp16 layer cases in a process of its own."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, **{env[0]: env[1]})
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_layers.py"), "-x", "-q", "-k",
                        "(test_conv_mfma or test_conv_epilogue_variants or test_conv_batch256) and not fp32"],
                       env=e, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
