"""Kernel-level parity of the non-GEMM kernels -- se_pool / se_fc / se_scale and head_tail (csrc/hip/small_ops.h) --
against the CPU oracle's restatements of the reference layers they replace:
GlobalPooling<false/true> (src/neural/blas/se_unit.cc:9-68), FullyConnect (fullyconnect.cc:7-19), SEUnit::Forward
(se_unit.cc:70-128) and the head tail of BlasForwardPipe::Forward (blas_forward_pipe.cc:496-580), through the oracle's
layer taps (oracle/sayuri_oracle.c so_tap_*).  Board sizes 2..19, mixed batches, all eight activations.

fp32 engine: abs <= 2e-5 * scale.  fp16 engine: the kernels see fp16-rounded activations (the oracle is given the same
rounded values) and store fp16 -> 2e-3 * scale on activations, 1e-4 on the fp32 outputs of the heads / gates."""
import ctypes

import numpy as np
import pytest

from _oracle import PortNet
from sayuri_amd import _lib

pytestmark = pytest.mark.gpu

FP = ctypes.POINTER(ctypes.c_float)


def _fp(a):
    return a.ctypes.data_as(FP)


def oracle():
    lib = PortNet.lib()
    lib.so_tap_se_unit.argtypes = [ctypes.c_int] * 3 + [FP] * 6 + [ctypes.c_int]
    lib.so_tap_global_pool.argtypes = [ctypes.c_int, ctypes.c_int, FP, FP, ctypes.c_int]
    lib.so_tap_fully_connect.argtypes = [ctypes.c_int, ctypes.c_int, FP, FP, FP, FP, ctypes.c_int]
    lib.so_tap_head_tail.argtypes = [ctypes.c_int] * 7 + [FP] * 18
    return lib


def r16(a, fp16):
    return a.astype(np.float16).astype(np.float32) if fp16 else a


BOARDS = [[19], [9, 13, 19], [2, 3, 5, 7, 19, 4], [19] * 5, [13, 13, 9, 9, 9, 19, 6]]


@pytest.mark.parametrize("fp16", [False, True], ids=["fp32", "fp16"])
@pytest.mark.parametrize("act", range(8))
def test_se_unit_kernels(act, fp16):
    lib, o = _lib.hip(), oracle()
    rng = np.random.default_rng(100 + act)
    for bsz, C, se, with_res in ((BOARDS[act % len(BOARDS)], 64, 16, True), (BOARDS[(act + 1) % len(BOARDS)], 96, 24, False),
                                 ([19, 9], 256, 64, True)):
        n = len(bsz)
        xs = [r16(rng.standard_normal((C, b * b)).astype(np.float32), fp16) for b in bsz]
        rs = [r16(rng.standard_normal((C, b * b)).astype(np.float32), fp16) for b in bsz] if with_res else None
        w1 = (rng.standard_normal((se, 3 * C)) / np.sqrt(3 * C)).astype(np.float32)
        b1 = (rng.standard_normal(se) * 0.1).astype(np.float32)
        w2 = (rng.standard_normal((2 * C, se)) / np.sqrt(se)).astype(np.float32)
        b2 = (rng.standard_normal(2 * C) * 0.1).astype(np.float32)
        xcat = np.concatenate([x.ravel() for x in xs])
        rcat = np.concatenate([r.ravel() for r in rs]) if rs else None
        y = np.zeros_like(xcat)
        gate = np.zeros((n, 2 * C), np.float32)
        bs_arr = np.asarray(bsz, np.int32)
        lib.sayuri_hip_test_se_unit.argtypes = [ctypes.c_int] * 3 + [_lib.c_int_p] + [ctypes.c_int] * 4 + [FP] * 8
        rc = lib.sayuri_hip_test_se_unit(0, int(fp16), n, bs_arr.ctypes.data_as(_lib.c_int_p), 19, C, se, act, _fp(xcat),
                                         _fp(rcat) if rs else None, _fp(w1), _fp(b1), _fp(w2), _fp(b2), _fp(y), _fp(gate))
        assert rc == 0, lib.sayuri_hip_last_error().decode()
        off = 0
        for i, b in enumerate(bsz):
            S = b * b
            # GlobalPooling<false> + both FullyConnects -> the gate
            pool = np.zeros(3 * C, np.float32)
            o.so_tap_global_pool(b, C, _fp(xs[i]), _fp(pool), 0)
            mid = np.zeros(se, np.float32)
            o.so_tap_fully_connect(3 * C, se, _fp(w1), _fp(b1), _fp(pool), _fp(mid), act)
            exc = np.zeros(2 * C, np.float32)
            o.so_tap_fully_connect(se, 2 * C, _fp(w2), _fp(b2), _fp(mid), _fp(exc), 0)
            exp_gate = np.concatenate([1.0 / (1.0 + np.exp(-exc[:C].astype(np.float64))), exc[C:]])
            assert np.abs(gate[i] - exp_gate).max() <= 1e-4 * max(1.0, np.abs(exp_gate).max()), (bsz, i, "gate")
            # the whole unit
            ref = xs[i].copy()
            o.so_tap_se_unit(b, C, se, _fp(w1), _fp(b1), _fp(w2), _fp(b2), _fp(ref), _fp(rs[i]) if rs else None, act)
            got = y[off:off + C * S].reshape(C, S)
            off += C * S
            scale = max(1.0, float(np.abs(ref).max()))
            tol = (2e-3 if fp16 else 2e-5) * scale
            assert np.isfinite(got).all()
            assert np.abs(got - ref).max() <= tol, (bsz, i, act, float(np.abs(got - ref).max()), tol)


@pytest.mark.parametrize("fp16", [False, True], ids=["fp32", "fp16"])
@pytest.mark.parametrize("act", range(8))
def test_head_tail_kernel(act, fp16):
    lib, o = _lib.hip(), oracle()
    rng = np.random.default_rng(200 + act)
    for bsz, Cp, Cv in ((BOARDS[act % len(BOARDS)], 24, 24), (BOARDS[(act + 2) % len(BOARDS)], 32, 48), ([19, 13], 48, 32)):
        n, prob_ch, pass_outs, misc_outs, B2 = len(bsz), 5, 5, 15, 361
        pcs = [r16(rng.standard_normal((Cp, b * b)).astype(np.float32), fp16) for b in bsz]
        vcs = [r16(rng.standard_normal((Cv, b * b)).astype(np.float32), fp16) for b in bsz]
        shapes = [(Cp, 3 * Cp), (Cp,), (pass_outs, Cp), (pass_outs,), (3 * Cv, 3 * Cv), (3 * Cv,), (misc_outs, 3 * Cv), (misc_outs,),
                  (prob_ch, Cp), (prob_ch,), (Cv,), (1,)]
        ws = [(rng.standard_normal(s) / np.sqrt(s[-1] if len(s) > 1 else 4)).astype(np.float32) for s in shapes]
        warr = (FP * 12)(*[_fp(w) for w in ws])
        pcat = np.concatenate([p.ravel() for p in pcs])
        vcat = np.concatenate([v.ravel() for v in vcs])
        prob = np.zeros((n, prob_ch, B2), np.float32)
        pas = np.zeros((n, pass_outs), np.float32)
        misc = np.zeros((n, misc_outs), np.float32)
        own = np.zeros((n, B2), np.float32)
        bs_arr = np.asarray(bsz, np.int32)
        lib.sayuri_hip_test_head_tail.argtypes = [ctypes.c_int] * 3 + [_lib.c_int_p] + [ctypes.c_int] * 7 + [FP, FP, ctypes.POINTER(FP)] + [FP] * 4
        rc = lib.sayuri_hip_test_head_tail(0, int(fp16), n, bs_arr.ctypes.data_as(_lib.c_int_p), 19, Cp, Cv, prob_ch, pass_outs, misc_outs, act,
                                           _fp(pcat), _fp(vcat), warr, _fp(prob), _fp(pas), _fp(misc), _fp(own))
        assert rc == 0, lib.sayuri_hip_last_error().decode()
        for i, b in enumerate(bsz):
            S = b * b
            e_prob, e_pass = np.zeros((prob_ch, S), np.float32), np.zeros(pass_outs, np.float32)
            e_own, e_misc = np.zeros(S, np.float32), np.zeros(misc_outs, np.float32)
            pc = pcs[i].copy()
            o.so_tap_head_tail(b, Cp, Cv, prob_ch, pass_outs, misc_outs, act, _fp(pc), _fp(vcs[i]), *[_fp(w) for w in ws],
                               _fp(e_prob), _fp(e_pass), _fp(e_own), _fp(e_misc))
            tol = 2e-4
            got_prob = prob[i].reshape(prob_ch, 19, 19)[:, :b, :b].reshape(prob_ch, S)
            got_own = own[i].reshape(19, 19)[:b, :b].ravel()
            assert np.abs(got_prob - e_prob).max() <= tol * max(1.0, np.abs(e_prob).max()), (bsz, i, "prob")
            assert np.abs(got_own - e_own).max() <= tol * max(1.0, np.abs(e_own).max()), (bsz, i, "own")
            assert np.abs(pas[i] - e_pass).max() <= tol * max(1.0, np.abs(e_pass).max()), (bsz, i, "pass")
            assert np.abs(misc[i] - e_misc).max() <= tol * max(1.0, np.abs(e_misc).max()), (bsz, i, "misc")
            # off-board cells of a smaller sample stay 0 in the NN grid
            mask = np.ones((19, 19), bool)
            mask[:b, :b] = False
            assert not prob[i].reshape(prob_ch, 19, 19)[:, mask].any() and not own[i].reshape(19, 19)[mask].any()


def conv3x3_f64(x, w, bias, b):
    """float64 direct 3x3 convolution of one sample: x [C][b*b], w [K][C][3][3] -> [K][b*b]"""
    C, K = x.shape[0], w.shape[0]
    xp = np.zeros((C, b + 2, b + 2), np.float64)
    xp[:, 1:-1, 1:-1] = x.reshape(C, b, b)
    y = np.zeros((K, b, b), np.float64)
    for dy in range(3):
        for dx in range(3):
            y += np.einsum("kc,cyx->kyx", w[:, :, dy, dx].astype(np.float64), xp[:, dy:dy + b, dx:dx + b])
    return (y + bias.astype(np.float64)[:, None, None]).reshape(K, b * b)


@pytest.mark.parametrize("via_tower", [0, 1], ids=["per-layer kernel", "tower kernel"])
@pytest.mark.parametrize("act", range(8))
def test_conv_with_se_unit_inside(act, via_tower):
    """conv_board_se_kernel / the SE body of the persistent tower kernel at kernel level: float64 convolution followed by
    the oracle's SEUnit::Forward tap (se_unit.cc:70-128) on it.  Boards 2..19, C = 96 / 128 / 256, se = 24 / 32 / 64,
    one sample per tile (fused) and several samples per tile (the tap must report the fallback)."""
    lib, o = _lib.hip(), oracle()
    rng = np.random.default_rng(300 + act)
    lib.sayuri_hip_test_conv_se.argtypes = [ctypes.c_int] * 2 + [_lib.c_int_p] + [ctypes.c_int] * 5 + [FP] * 9
    cases = [([19, 19, 19], 256, 64, True), ([19] * 2, 128, 32, False), ([19, 17, 16, 15, 14], 128, 24, True), ([19], 96, 24, True)]
    # small boards one per batch (a batch of several small boards shares a tile: checked below)
    cases += [([b], 128, 32, bool(b & 1)) for b in (2, 3, 5, 9, 13)][act % 5:act % 5 + 2]
    for bsz, C, se, with_res in cases:
        n = len(bsz)
        xs = [r16(rng.standard_normal((C, b * b)).astype(np.float32), True) for b in bsz]
        rs = [r16(rng.standard_normal((C, b * b)).astype(np.float32), True) for b in bsz] if with_res else None
        w = r16((rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32), True)
        bias = (rng.standard_normal(C) * 0.1).astype(np.float32)
        w1 = (rng.standard_normal((se, 3 * C)) / np.sqrt(3 * C)).astype(np.float32)
        b1 = (rng.standard_normal(se) * 0.1).astype(np.float32)
        w2 = (rng.standard_normal((2 * C, se)) / np.sqrt(se)).astype(np.float32)
        b2 = (rng.standard_normal(2 * C) * 0.1).astype(np.float32)
        xcat = np.concatenate([x.ravel() for x in xs])
        rcat = np.concatenate([r.ravel() for r in rs]) if rs else None
        y = np.zeros_like(xcat)
        bs_arr = np.asarray(bsz, np.int32)
        rc = lib.sayuri_hip_test_conv_se(0, n, bs_arr.ctypes.data_as(_lib.c_int_p), 19, C, se, act, via_tower, _fp(xcat), _fp(w), _fp(bias),
                                         _fp(rcat) if rs else None, _fp(w1), _fp(b1), _fp(w2), _fp(b2), _fp(y))
        if C % 128:  # no board kernel for this channel count (the engine runs such layers on conv_mfma + the separate SE kernels)
            assert rc == 1
            continue
        assert rc == 0, (bsz, C, rc, lib.sayuri_hip_last_error().decode())
        off = 0
        for i, b in enumerate(bsz):
            S = b * b
            ref = conv3x3_f64(xs[i], w, bias, b).astype(np.float32)
            ref = np.ascontiguousarray(ref)
            o.so_tap_se_unit(b, C, se, _fp(w1), _fp(b1), _fp(w2), _fp(b2), _fp(ref), _fp(rs[i]) if rs else None, act)
            got = y[off:off + C * S].reshape(C, S)
            off += C * S
            scale = max(1.0, float(np.abs(ref).max()))
            assert np.isfinite(got).all()
            # fp16 store + fp16 FC weights: 3e-3 of the output scale
            assert np.abs(got - ref).max() <= 3e-3 * scale, (bsz, i, C, act, float(np.abs(got - ref).max()), scale)
    # several samples in one tile: the fused kernel does not apply, the tap says so (the engine runs conv + se_pool / se_fc / se_scale)
    bs_arr = np.asarray([9, 9, 9, 9], np.int32)
    dummy = np.zeros(4 * 128 * 81, np.float32)
    w = np.zeros((128, 128, 3, 3), np.float32)
    z = np.zeros(3 * 128 * 32 + 512, np.float32)
    rc = lib.sayuri_hip_test_conv_se(0, 4, bs_arr.ctypes.data_as(_lib.c_int_p), 19, 128, 32, act, via_tower, _fp(dummy), _fp(w), _fp(z), None,
                                     _fp(z), _fp(z), _fp(z), _fp(z), _fp(dummy.copy()))
    assert rc == 1


@pytest.mark.parametrize("act", range(8))
def test_head_board_kernel(act):
    """head_board_kernel at kernel level: the two 1x1 head convolutions in float64 (on the fp16-rounded trunk), rounded to
    nothing, then the oracle's head-tail tap (blas_forward_pipe.cc:449-580).  Boards 2..19, C = 128 / 256, Cp / Cv =
    24 / 32 / 48."""
    lib, o = _lib.hip(), oracle()
    rng = np.random.default_rng(400 + act)
    lib.sayuri_hip_test_head_board.argtypes = [ctypes.c_int] * 2 + [_lib.c_int_p] + [ctypes.c_int] * 8 + [FP] * 5 + [ctypes.POINTER(FP)] + [FP] * 4
    for bsz, C, Cp, Cv in ((BOARDS[act % len(BOARDS)], 128, 24, 24), (BOARDS[(act + 2) % len(BOARDS)], 256, 32, 32), ([19, 13, 2], 256, 32, 48),
                           ([19, 5], 128, 48, 32)):
        n, prob_ch, pass_outs, misc_outs, B2 = len(bsz), 5, 5, 15, 361
        ts = [r16(rng.standard_normal((C, b * b)).astype(np.float32), True) for b in bsz]
        p_w = r16((rng.standard_normal((Cp, C)) / np.sqrt(C)).astype(np.float32), True)
        v_w = r16((rng.standard_normal((Cv, C)) / np.sqrt(C)).astype(np.float32), True)
        p_b = (rng.standard_normal(Cp) * 0.1).astype(np.float32)
        v_b = (rng.standard_normal(Cv) * 0.1).astype(np.float32)
        shapes = [(Cp, 3 * Cp), (Cp,), (pass_outs, Cp), (pass_outs,), (3 * Cv, 3 * Cv), (3 * Cv,), (misc_outs, 3 * Cv), (misc_outs,),
                  (prob_ch, Cp), (prob_ch,), (Cv,), (1,)]
        ws = [(rng.standard_normal(s) / np.sqrt(s[-1] if len(s) > 1 else 4)).astype(np.float32) for s in shapes]
        ws[8] = r16(ws[8], True)   # the per-pixel weights are an fp16 MFMA image in the kernel
        ws[10] = r16(ws[10], True)
        warr = (FP * 12)(*[_fp(w) for w in ws])
        tcat = np.concatenate([t.ravel() for t in ts])
        prob = np.zeros((n, prob_ch, B2), np.float32)
        pas = np.zeros((n, pass_outs), np.float32)
        misc = np.zeros((n, misc_outs), np.float32)
        own = np.zeros((n, B2), np.float32)
        bs_arr = np.asarray(bsz, np.int32)
        rc = lib.sayuri_hip_test_head_board(0, n, bs_arr.ctypes.data_as(_lib.c_int_p), 19, C, Cp, Cv, prob_ch, pass_outs, misc_outs, act,
                                            _fp(tcat), _fp(p_w), _fp(p_b), _fp(v_w), _fp(v_b), warr, _fp(prob), _fp(pas), _fp(misc), _fp(own))
        assert rc == 0, (bsz, C, Cp, Cv, rc, lib.sayuri_hip_last_error().decode())
        for i, b in enumerate(bsz):
            S = b * b
            from test_gpu_layers import act_np
            pc = act_np(p_w.astype(np.float64) @ ts[i].astype(np.float64) + p_b[:, None], act).astype(np.float32)
            vc = act_np(v_w.astype(np.float64) @ ts[i].astype(np.float64) + v_b[:, None], act).astype(np.float32)
            pc, vc = np.ascontiguousarray(pc), np.ascontiguousarray(vc)
            e_prob, e_pass = np.zeros((prob_ch, S), np.float32), np.zeros(pass_outs, np.float32)
            e_own, e_misc = np.zeros(S, np.float32), np.zeros(misc_outs, np.float32)
            o.so_tap_head_tail(b, Cp, Cv, prob_ch, pass_outs, misc_outs, act, _fp(pc), _fp(vc), *[_fp(w) for w in ws],
                               _fp(e_prob), _fp(e_pass), _fp(e_own), _fp(e_misc))
            # the head planes stay in fp32 registers; the per-pixel product runs on fp16-rounded planes: 2e-3 of the scale
            tol = 2e-3
            got_prob = prob[i].reshape(prob_ch, 19, 19)[:, :b, :b].reshape(prob_ch, S)
            got_own = own[i].reshape(19, 19)[:b, :b].ravel()
            assert np.abs(got_prob - e_prob).max() <= tol * max(1.0, np.abs(e_prob).max()), (bsz, i, "prob", float(np.abs(got_prob - e_prob).max()))
            assert np.abs(got_own - e_own).max() <= tol * max(1.0, np.abs(e_own).max()), (bsz, i, "own")
            assert np.abs(pas[i] - e_pass).max() <= tol * max(1.0, np.abs(e_pass).max()), (bsz, i, "pass")
            assert np.abs(misc[i] - e_misc).max() <= tol * max(1.0, np.abs(e_misc).max()), (bsz, i, "misc")
            mask = np.ones((19, 19), bool)
            mask[:b, :b] = False
            assert not prob[i].reshape(prob_ch, 19, 19)[:, mask].any() and not own[i].reshape(19, 19)[mask].any()
