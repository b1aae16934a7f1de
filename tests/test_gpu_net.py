"""Whole-network parity of the HIP pipe against the CPU oracle and the reference goldens.

Tolerances (raw, pre-softmax outputs; magnitudes are O(1)):
  fp32 engine : abs <= 1e-4   (SURVEY.md 8c gate; north_star "within fp32 tolerance")
  fp16 engine : abs <= 4e-3 * max(1, max|expected|) -- twice the largest error measured over every golden fixture
                against the reference (1.9e-3 of the output scale on tiny_all, 5.9e-4 = 2.3e-3 abs on 64 positions of
                the 20b256 network; profiles/r02_fp16_error_fixtures.txt, profiles/r02_fp16_error_20b256.json,
                test_fp16_error_is_measured_on_64_positions) -- and the reference's own GPU-vs-CPU SelfCheck
                criterion L2(softmax policy ++ pass ++ wdl_winrate) <= 0.2 (network.cc:333-359) as the hard floor.
"""
import ctypes

import numpy as np
import pytest

from _golden import Golden
from _oracle import PortNet
from golden_specs import FIXTURES
from sayuri_amd import weights as W
from sayuri_amd.pipe import HipForwardPipe

pytestmark = pytest.mark.gpu

FP32_ATOL = 1e-4
FP16_ATOL = 4e-3  # times max(1, output scale): fp16_tol()


def fp16_tol(exp):
    return FP16_ATOL * max(1.0, float(np.abs(exp).max()))


def self_check_l2(got, exp, bs):
    """reference Network::SelfCheck (network.cc:333-359) on post-processed outputs."""
    a, b = PortNet.postprocess(got, bs), PortNet.postprocess(exp, bs)
    s = bs * bs
    va = np.concatenate([a[:s + 1], [a[2 * s + 1 + 3]]])
    vb = np.concatenate([b[:s + 1], [b[2 * s + 1 + 3]]])
    return float(np.sqrt(((va - vb) ** 2).sum()))


def check(pipe, cases, atol, label):
    planes = [c[0] for c in cases]
    bsz = [c[1] for c in cases]
    offs = [c[2] for c in cases]
    for mode, outs in (("batch", pipe.BatchForward(planes, bsz, offsets=offs)),
                       ("queue", pipe.Forward(planes, bsz, offsets=offs))):
        for (p, bs, off, exp), got in zip(cases, outs):
            assert got.shape == exp.shape
            assert np.isfinite(got).all(), (label, mode)
            err = float(np.abs(got - exp).max())
            assert err <= (atol(exp) if callable(atol) else atol), (label, mode, bs, off, err)
            assert self_check_l2(got, exp, bs) <= 0.2


@pytest.mark.parametrize("fp16", [False, True], ids=["fp32", "fp16"])
@pytest.mark.parametrize("name", [fx["name"] for fx in FIXTURES if fx["name"].startswith("tiny")] + ["net_6b96"])
def test_golden_parity(name, fp16, tmp_weights_dir):
    """HIP pipe vs the outputs of the reference's own BlasForwardPipe (tests/golden)."""
    g = Golden(name, tmp_weights_dir)
    cases = [(g.planes(c), c["board_size"], c["offset"], g.expected(c)) for c in g.cases if c["winograd"] == 1]
    pipe = HipForwardPipe(g.weights_path, board_size=19, batch_size=16, fp16=fp16)
    try:
        check(pipe, cases, fp16_tol if fp16 else FP32_ATOL, name)
    finally:
        pipe.Destroy()


@pytest.mark.parametrize("variant", ["glds", "v0"])
def test_20b256_fallback_conv_paths(variant, tmp_weights_dir, monkeypatch):
    """The same network with the one-workgroup-per-board convolution switched off: SAYURI_CONV=glds (LDS-DMA tiles across
    samples, conv_glds.h -- what batches of boards that do not fit a board tile use) and v0 (the generic kernel)."""
    monkeypatch.setenv("SAYURI_CONV", variant)
    g = Golden("net_20b256", tmp_weights_dir)
    cases = [(g.planes(c), c["board_size"], c["offset"], g.expected(c)) for c in g.cases]
    pipe = HipForwardPipe(g.weights_path, board_size=19, batch_size=8, fp16=True)
    try:
        check(pipe, cases, fp16_tol, "20b256-" + variant)
    finally:
        pipe.Destroy()


@pytest.mark.parametrize("fp16", [False, True], ids=["fp32", "fp16"])
def test_20b256_golden_and_oracle(fp16, tmp_weights_dir):
    """BASELINE.json configs[1] network: golden cases + fresh oracle evaluations."""
    g = Golden("net_20b256", tmp_weights_dir)
    cases = [(g.planes(c), c["board_size"], c["offset"], g.expected(c)) for c in g.cases]
    oracle = PortNet(g.weights_path)
    for i, bs in enumerate((19, 9)):
        p = W.synthetic_planes(1, bs, seed=900 + i)[0]
        cases.append((p, bs, i, oracle.forward(p, bs, offset=i)))
    pipe = HipForwardPipe(g.weights_path, board_size=19, batch_size=8, fp16=fp16)
    try:
        check(pipe, cases, fp16_tol if fp16 else FP32_ATOL, "20b256")
    finally:
        pipe.Destroy()


def test_fp16_storage_range_on_the_hot_network(tmp_weights_dir):
    """fp16 stress fixture: the 20b x 256 architecture with BN statistics that let the residual stream
    grow to |x| ~ 1.4e3 by block 20 and the raw outputs to ~1e3 (tests/golden_specs.py net_20b256_hot; goldens from the
    reference's BlasForwardPipe).  The fp16 engine stores 41 layers of such activations as fp16 (max 65 504, 11 bits):
    outputs must stay finite, within the gate RELATIVE to the output scale, and inside the reference's own GPU-vs-CPU
    criterion (Network::SelfCheck, network.cc:333-359: L2 <= 0.2).  The measured error goes to gpurun_out/ for profiles/."""
    import json
    import os
    g = Golden("net_20b256_hot", tmp_weights_dir)
    cases = [(g.planes(c), c["board_size"], c["offset"], g.expected(c)) for c in g.cases]
    rows = []
    for fp16 in (True, False):
        pipe = HipForwardPipe(g.weights_path, board_size=19, batch_size=8, fp16=fp16)
        try:
            outs = pipe.BatchForward([c[0] for c in cases], [c[1] for c in cases], offsets=[c[2] for c in cases])
        finally:
            pipe.Destroy()
        for (p, bs, off, exp), got in zip(cases, outs):
            scale = float(np.abs(exp).max())
            err = float(np.abs(got - exp).max())
            l2 = self_check_l2(got, exp, bs)
            rows.append({"fp16": fp16, "output_scale": scale, "max_abs_err": err, "rel_to_scale": err / scale, "selfcheck_l2": l2})
            assert np.isfinite(got).all(), fp16
            assert scale > 100.0
            assert err <= (FP16_ATOL if fp16 else FP32_ATOL) * scale, (fp16, err, scale)
            assert l2 <= 0.2, (fp16, l2)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "fp16_error_hot_network.json"), "w") as f:
        json.dump({"fixture": "net_20b256_hot (branch_scale 5.5, seed 24): residual stream max |x| ~ 1.4e3 at block 20",
                   "worst_fp16_rel_to_scale": max(r["rel_to_scale"] for r in rows if r["fp16"]),
                   "worst_fp16_selfcheck_l2": max(r["selfcheck_l2"] for r in rows if r["fp16"]), "cases": rows}, f, indent=1)


def test_mixed_board_batch_matches_native_evaluation(tmp_weights_dir):
    """configs[4] mechanism: 9/13/19 samples in one batch on a 19x19 graph equal the oracle's
    native small-board evaluation of each sample (no mask error, fp32 engine)."""
    g = Golden("tiny_all", tmp_weights_dir)
    oracle = PortNet(g.weights_path)
    rng = np.random.default_rng(3)
    bsz = [int(b) for b in rng.choice([9, 13, 19], size=24)] + [7, 2, 19]
    planes = W.synthetic_planes(len(bsz), bsz, seed=77)
    pipe = HipForwardPipe(g.weights_path, board_size=19, batch_size=32, fp16=False)
    try:
        outs = pipe.BatchForward(planes, bsz)
        for p, bs, got in zip(planes, bsz, outs):
            exp = oracle.forward(p, bs)
            assert np.abs(got - exp).max() <= FP32_ATOL, bs
    finally:
        pipe.Destroy()


@pytest.mark.parametrize("seed", [1, 2, 7])
@pytest.mark.parametrize("fp16", [False, True], ids=["fp32", "fp16"])
def test_full_mixed_batches_fit_a_tile_configuration(seed, fp16, tmp_weights_dir):
    """256 samples of randomly mixed 9/13/19 boards (what a mixed-size self-play queue produces): pixel tiles that
    straddle several small boards need up to 1.6x their pixel count in halo positions -- every layer, including the
    1x1 head convolutions of the generic kernel, must still find a tile configuration (seed 1 used to fail with
    "no conv tile configuration fits this batch geometry").  Spot-checked against the oracle."""
    g = Golden("net_6b96", tmp_weights_dir)
    oracle = PortNet(g.weights_path)
    rng = np.random.default_rng(seed)
    bsz = [int(b) for b in rng.choice([9, 13, 19], size=256)]
    planes = W.synthetic_planes(len(bsz), bsz, seed=seed)
    pipe = HipForwardPipe(g.weights_path, board_size=19, batch_size=256, fp16=fp16)
    try:
        outs = pipe.BatchForward(planes, bsz)
        for i in (0, 17, 101, 255):
            exp = oracle.forward(planes[i], bsz[i])
            assert np.abs(outs[i] - exp).max() <= (fp16_tol(exp) if fp16 else FP32_ATOL), (i, bsz[i])
    finally:
        pipe.Destroy()


def test_batch256_properties_20b256(tmp_weights_dir):
    """Full bench size (batch 256, 19x19, 20b256, fp16): size-independent properties --
    a sample's result does not depend on its slot or on its neighbours, duplicated inputs give
    bit-identical outputs, and sixteen slots (sixteen different positions, each in its own workgroup of the persistent
    tower launch) agree with the oracle."""
    g = Golden("net_20b256", tmp_weights_dir)
    n, nb = 256, 16
    base = W.synthetic_planes(nb, 19, seed=1234)
    idx = np.arange(n) % nb
    planes = [base[i] for i in idx]
    pipe = HipForwardPipe(g.weights_path, board_size=19, batch_size=n, fp16=True)
    try:
        outs = pipe.BatchForward(planes, [19] * n)
        for i in range(nb, n):
            np.testing.assert_array_equal(outs[i], outs[i % nb])
        rev = pipe.BatchForward(planes[::-1], [19] * n)
        for i in range(n):
            np.testing.assert_array_equal(rev[i], outs[n - 1 - i])
        oracle = PortNet(g.weights_path)
        for i in range(nb):
            exp = oracle.forward(base[i], 19)
            assert np.abs(outs[i + 7 * nb] - exp).max() <= fp16_tol(exp), i
        one = pipe.BatchForward([base[3]], [19])[0]
        assert np.abs(one - outs[3]).max() <= 1e-6  # batch of 1 vs inside a batch of 256
    finally:
        pipe.Destroy()


def test_reconstruct_smaller_board(tmp_weights_dir):
    """Network::Reconstruct path: NN board 9 graph evaluates 9x9 natively (BASELINE configs[0] net)."""
    g = Golden("net_6b96", tmp_weights_dir)
    oracle = PortNet(g.weights_path)
    pipe = HipForwardPipe(g.weights_path, board_size=19, batch_size=4, fp16=False)
    try:
        pipe.Construct(9, 4)
        p = W.synthetic_planes(3, 9, seed=5)
        outs = pipe.BatchForward(p, [9, 9, 9])
        for x, got in zip(p, outs):
            assert np.abs(got - oracle.forward(x, 9)).max() <= FP32_ATOL
    finally:
        pipe.Destroy()


def test_se_unit_fused_into_the_convolution_matches_the_separate_kernels(tmp_weights_dir, monkeypatch):
    """20b256 has a squeeze-and-excitation unit on every third block.  With one sample per board tile the unit runs
    inside the block's second convolution (conv_board.h: pooling over the accumulators, both FCs, the gate); with
    SAYURI_SE_FUSED=0 it runs as se_pool / se_fc / se_scale on the fp16 activations.  Both against the oracle, and against
    each other: they differ only by where x is rounded to fp16."""
    g = Golden("net_20b256", tmp_weights_dir)
    oracle = PortNet(g.weights_path)
    bsz = [19, 13, 19, 9, 19, 19, 7, 19, 16, 14]  # full boards and boards that can share a tile, in one batch
    planes = W.synthetic_planes(len(bsz), bsz, seed=4242)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SAYURI_SE_FUSED", mode)
        pipe = HipForwardPipe(g.weights_path, board_size=19, batch_size=16, fp16=True)
        try:
            outs[mode] = pipe.BatchForward(planes, bsz)
        finally:
            pipe.Destroy()
    for i, (p, bs) in enumerate(zip(planes, bsz)):
        exp = oracle.forward(p, bs)
        assert np.abs(outs["1"][i] - exp).max() <= fp16_tol(exp), (i, bs)
        assert np.abs(outs["0"][i] - exp).max() <= fp16_tol(exp), (i, bs)
        assert np.abs(outs["1"][i] - outs["0"][i]).max() <= fp16_tol(exp)
        # which samples the fused form takes is decided by their board size alone (Engine::conv_se: a board that can never
        # share a tile, bs >= 14): for those the switch changes the path, the smaller boards go through the separate kernels
        # either way -- bit for bit the same
        if 2 * bs * bs > 384:
            assert np.abs(outs["1"][i] - outs["0"][i]).max() > 0, "the switch did not change the path"
        else:
            assert np.array_equal(outs["1"][i], outs["0"][i]), (i, bs)


@pytest.mark.parametrize("name", ["net_20b256", "net_40b384", "net_6b96", "tiny_res", "tiny_all"])
def test_heads_fused_kernel_matches_the_separate_head_kernels(name, tmp_weights_dir, monkeypatch):
    """fp16 engine, normal policy head: both 1x1 head convolutions, the pooling, the four FCs and the per-pixel planes run
    as one workgroup per sample (head_board.h); SAYURI_HEADS_FUSED=0 runs conv1x1 x2 + head_tail_kernel.  Both against the
    oracle, mixed board sizes in caller order (the fused kernel writes through the sort permutation)."""
    g = Golden(name, tmp_weights_dir)
    oracle = PortNet(g.weights_path)
    bsz = [19, 13, 19, 9, 19, 19, 7, 19, 13]
    planes = W.synthetic_planes(len(bsz), bsz, seed=777)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SAYURI_HEADS_FUSED", mode)
        pipe = HipForwardPipe(g.weights_path, board_size=19, batch_size=16, fp16=True)
        try:
            outs[mode] = pipe.BatchForward(planes, bsz)
        finally:
            pipe.Destroy()
    for i, (p, bs) in enumerate(zip(planes, bsz)):
        exp = oracle.forward(p, bs)
        assert np.abs(outs["1"][i] - exp).max() <= fp16_tol(exp), (i, bs)
        assert np.abs(outs["0"][i] - exp).max() <= fp16_tol(exp), (i, bs)
    if name != "tiny_all":  # tiny_all has a RepLK policy head: the fused kernel does not apply, both runs are the same path
        assert any(np.abs(outs["1"][i] - outs["0"][i]).max() > 0 for i in range(len(bsz))), "the switch did not change the path"


@pytest.mark.parametrize("name,fp16", [("net_20b256", True), ("tiny_res", False), ("tiny_all", True)])
def test_packed_planes_give_identical_outputs(name, fp16, tmp_weights_dir):
    """the planes as bit planes + broadcast scalars (csrc/host/packed_planes.h, 1.8 KB instead of 62 KB per
    sample) expanded by pack_bits_kernel give the network the same activations as the fp32 planes through pack_input_kernel:
    the raw outputs of sayuri_hip_forward_packed and sayuri_hip_forward are bit-identical (mixed board sizes, caller order),
    and so are ForwardPacked / Forward through the queue, all-packed and mixed batches."""
    from sayuri_amd.engine import pack_planes
    from sayuri_amd.pipe import hip_forward_packed_raw, hip_forward_raw
    g = Golden(name, tmp_weights_dir)
    bsz = [19, 9, 13, 19, 7, 19, 13, 19, 19, 9, 19]
    planes = W.synthetic_planes(len(bsz), bsz, seed=31337)
    for p in planes:  # rule and wave planes away from 0 too
        p[37] = 1.0
        p[38] = 0.25
    B = 19
    grid = np.zeros((len(bsz), 43, B * B), np.float32)
    for i, (p, bs) in enumerate(zip(planes, bsz)):
        grid[i].reshape(43, B, B)[:, :bs, :bs] = p.reshape(43, bs, bs)
    records = np.stack([pack_planes(p, 37) for p in planes])
    pipe = HipForwardPipe(g.weights_path, board_size=B, batch_size=16, fp16=fp16)
    try:
        ctx = pipe.ctx(0)
        a = hip_forward_raw(ctx, grid, bsz, B)
        b = hip_forward_packed_raw(ctx, records, 37, bsz, B)
        for x, y, what in zip(a, b, ("prob", "pass", "misc", "own")):
            assert np.array_equal(x, y), what
        assert np.abs(a[0]).max() > 0
        # through the queue the batches form as the callers arrive; no sample's result depends on its batch mates, so all of
        # them compare exactly between the three routes (and to the oracle within the gate)
        q_fp32 = pipe.Forward(planes, bsz)
        q_pack = pipe.ForwardPacked(planes, bsz)
        q_mix = pipe.ForwardPacked(planes, bsz, mixed=True)
        oracle = PortNet(g.weights_path)
        tol = fp16_tol if fp16 else (lambda e: FP32_ATOL)
        for i, bs in enumerate(bsz):
            if bs == B:
                exp = oracle.forward(planes[i], bs)
                what = lambda: (i, [float(np.abs(q[i] - exp).max()) for q in (q_fp32, q_pack, q_mix)], float(np.abs(q_fp32[i] - q_pack[i]).max()),
                                float(np.abs(q_fp32[i] - q_mix[i]).max()), pipe.pump_times())
                assert np.array_equal(q_fp32[i], q_pack[i]), what()
                assert np.array_equal(q_fp32[i], q_mix[i]), what()
            else:
                # a small board too: which kernels it meets depends on its own size only (test_a_position_does_not_depend_on_its_batch_mates)
                exp = oracle.forward(planes[i], bs)
                for q in (q_fp32, q_pack, q_mix):
                    assert np.abs(q[i] - exp).max() <= tol(exp), i
                assert np.array_equal(q_fp32[i], q_pack[i]) and np.array_equal(q_fp32[i], q_mix[i]), (i, bs)
        exp = oracle.forward(planes[0], bsz[0])
        assert np.abs(q_pack[0] - exp).max() <= tol(exp)
    finally:
        pipe.Destroy()


@pytest.mark.parametrize("name", ["net_20b256", "net_40b384"])
def test_a_position_does_not_depend_on_its_batch_mates(name, tmp_weights_dir, capsys):
    """Reference batch_forward_pipe.cc:15-33,48-68: a request's result is a function of the request.  Here the kernels a sample
    meets depend on the batch geometry (which samples share a tile, which kernel family a layer takes), so this pins it: one
    9x9, one 13x13 and one 19x19 position, each evaluated ALONE and inside four different mixed batches (at different places,
    with different mates, batches of every kernel plan), must come out BIT-equal in fp16.  What makes it hold: the SE unit's
    form (fused into the convolution or three separate kernels, which round at different points) is chosen by the sample's
    board size alone (Engine::conv_se), and the convolution kernels accumulate in one order whatever the tile plan."""
    from sayuri_amd.pipe import hip_forward_raw
    g = Golden(name, tmp_weights_dir)
    B = 19
    probes = {bs: W.synthetic_planes(1, bs, seed=4200 + bs)[0] for bs in (9, 13, 19)}
    rng = np.random.default_rng(77)

    def grid_of(planes, bsz):
        gr = np.zeros((len(bsz), 43, B * B), np.float32)
        for i, (p, bs) in enumerate(zip(planes, bsz)):
            gr[i].reshape(43, B, B)[:, :bs, :bs] = p.reshape(43, bs, bs)
        return gr

    def sample(out, i):
        return [np.array(t[i]) for t in out]

    pipe = HipForwardPipe(g.weights_path, board_size=B, batch_size=64, fp16=True)
    try:
        ctx = pipe.ctx(0)
        for bs, probe in probes.items():
            alone = sample(hip_forward_raw(ctx, grid_of([probe], [bs]), [bs], B), 0)
            assert np.abs(alone[0]).max() > 0
            mates = [
                [19] * 5 + [9] * 7 + [13] * 3,            # a bit of everything
                [9] * 40,                                 # small boards only: four per tile
                [19] * 30 + [13] * 3 + [9] * 2,           # mostly full boards
                [13] * 9 + [7] * 6 + [19] * 2 + [9] * 21 + [16] * 3,
            ]
            for k, sizes in enumerate(mates):
                sizes = list(sizes)
                rng.shuffle(sizes)
                at = int(rng.integers(0, len(sizes) + 1))
                sizes.insert(at, bs)
                planes = W.synthetic_planes(len(sizes), sizes, seed=900 + 10 * bs + k)
                planes[at] = probe
                got = sample(hip_forward_raw(ctx, grid_of(planes, sizes), sizes, B), at)
                for a, b, what in zip(alone, got, ("prob", "pass", "misc", "own")):
                    assert np.array_equal(a, b), (name, bs, k, what, float(np.abs(a - b).max()))
    finally:
        pipe.Destroy()


def test_computed_table_entries_match_the_tables(tmp_weights_dir, monkeypatch):
    """Uniform batches of one-sample tiles (BoardParams::arith): the board kernels compute position -> source row and
    pixel -> position / output row instead of reading the tables board_setup_kernel built.  SAYURI_NO_ARITH=1 reads the
    tables: bit-identical outputs, on 19x19 (one board per tile) and on 13x13 as the NN board (one board per tile too)."""
    g = Golden("net_20b256", tmp_weights_dir)
    for board, n in ((19, 160), (19, 40), (13, 24)):
        planes = W.synthetic_planes(n, board, seed=99 + board)
        outs = {}
        for mode in ("0", "1"):
            if mode == "1":
                monkeypatch.setenv("SAYURI_NO_ARITH", "1")
            else:
                monkeypatch.delenv("SAYURI_NO_ARITH", raising=False)
            pipe = HipForwardPipe(g.weights_path, board_size=board, batch_size=n, fp16=True)  # 160: the 256-channel tile, 40: two 128-channel tiles
            try:
                outs[mode] = pipe.BatchForward(planes, [board] * n)
            finally:
                pipe.Destroy()
        for a, b in zip(outs["0"], outs["1"]):
            assert np.array_equal(a, b)
        assert np.abs(outs["0"][0]).max() > 0


def test_persistent_tower_and_its_weight_hand_over_change_nothing(tmp_weights_dir, monkeypatch):
    """One launch per convolution (SAYURI_TOWER=0), the persistent tower launch without the weight hand-over between its layers
    (SAYURI_TOWER_CHAIN=0) and with it (the default) run the same compiled main loop, SE stage and epilogue: bit-identical
    outputs -- on the 20b x 256 network (SE units included) at 256 boards (256-channel tile) and 40 boards (two 128-channel
    tiles: no run), and on 13x13 as the NN board."""
    g = Golden("net_20b256", tmp_weights_dir)
    for board, n in ((19, 256), (19, 40), (13, 200)):
        planes = W.synthetic_planes(n, board, seed=7 + n)
        outs = {}
        for mode, env in (("layers", {"SAYURI_TOWER": "0"}), ("run", {"SAYURI_TOWER_CHAIN": "0"}), ("run+handover", {})):
            for k in ("SAYURI_TOWER", "SAYURI_TOWER_CHAIN"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            pipe = HipForwardPipe(g.weights_path, board_size=board, batch_size=n, fp16=True)
            try:
                outs[mode] = pipe.BatchForward(planes, [board] * n)
            finally:
                pipe.Destroy()
        for other in ("run", "run+handover"):
            for a, b in zip(outs["layers"], outs[other]):
                assert np.array_equal(a, b), (board, n, other)
        assert np.abs(outs["layers"][0]).max() > 0


@pytest.mark.parametrize("fp16", [False, True], ids=["fp32", "fp16"])
def test_40b384_golden_parity(fp16, tmp_weights_dir):
    """BASELINE.json configs[4] network (40 blocks x 384 filters) on 19 / 13 / 9 boards against the reference's own
    BlasForwardPipe outputs (tests/golden/net_40b384.npz, generated by tests/golden/make_golden.py)."""
    g = Golden("net_40b384", tmp_weights_dir)
    cases = [(g.planes(c), c["board_size"], c["offset"], g.expected(c)) for c in g.cases]
    pipe = HipForwardPipe(g.weights_path, board_size=19, batch_size=8, fp16=fp16)
    try:
        check(pipe, cases, fp16_tol if fp16 else FP32_ATOL, "40b384")
    finally:
        pipe.Destroy()


@pytest.mark.parametrize("fp16", [False, True], ids=["fp32", "fp16"])
def test_config5_full_mixed_batch_40b384(fp16, tmp_weights_dir):
    """configs[4] at its real size on one GPU: a full batch of 256 samples of mixed 9 / 13 / 19 boards through the
    40-block x 384-filter network (reference mechanism: batch_forward_pipe.cc:15-33,48-68 re-pad + cuda_forward_pipe.cc:636-682
    masks; here the compact per-sample layout).  The golden positions sit inside the batch and must come out as the
    reference computed them; other slots are spot-checked against the oracle."""
    g = Golden("net_40b384", tmp_weights_dir)
    oracle = PortNet(g.weights_path)
    rng = np.random.default_rng(55)
    bsz = [int(b) for b in rng.choice([9, 13, 19], size=256)]
    planes = W.synthetic_planes(len(bsz), bsz, seed=5500)
    gold = {}
    for k, c in enumerate(g.cases):  # plant the golden positions at scattered slots
        slot = 17 + 60 * k
        bsz[slot] = c["board_size"]
        planes[slot] = g.planes(c)
        gold[slot] = (c["offset"], g.expected(c))
    tol = (lambda e: fp16_tol(e)) if fp16 else (lambda e: FP32_ATOL)
    pipe = HipForwardPipe(g.weights_path, board_size=19, batch_size=256, fp16=fp16)
    try:
        offs = [gold[i][0] if i in gold else 0 for i in range(256)]
        outs = pipe.BatchForward(planes, bsz, offsets=offs)
        for slot, (off, exp) in gold.items():
            assert np.abs(outs[slot] - exp).max() <= tol(exp), ("golden", slot, bsz[slot])
        for i in (0, 255):
            exp = oracle.forward(planes[i], bsz[i])
            assert np.abs(outs[i] - exp).max() <= tol(exp), ("oracle", i, bsz[i])
        assert all(np.isfinite(o).all() for o in outs)
    finally:
        pipe.Destroy()


def test_split_se_unit_inside_the_convolution_40b384(tmp_weights_dir, monkeypatch):
    """Round 6: the SE unit of a layer whose channels are split over three 128-channel workgroups runs inside the convolution
    (conv_board_sx.h: pooling per sample through LDS, partial squeeze sums exchanged between the sibling workgroups, gate on the
    accumulators) instead of as se_pool / se_fc / se_scale.  A mixed batch -- one, two and four samples per tile, plus boards
    too small for the form (5x5, 7x7: the separate kernels behind the fused tiles) -- must (a) really take the kernel, (b) agree
    with the separate kernels (SAYURI_SE_SPLIT=0) within the fp16 gate (they round x to fp16 before the unit, the fused form does
    not), (c) match the oracle, (d) give a position the same BITS alone, in this batch and in a batch of other mates."""
    from sayuri_amd import _lib
    from sayuri_amd.pipe import hip_forward_raw
    g = Golden("net_40b384", tmp_weights_dir)
    net = PortNet(g.weights_path)
    B = 19
    rng = np.random.default_rng(66)
    bsz = [int(b) for b in rng.choice([9, 13, 19], size=90)] + [5, 7, 19, 7, 5, 13, 9]
    n = len(bsz)
    planes = W.synthetic_planes(n, bsz, seed=6600)
    grid = np.zeros((n, 43, B * B), np.float32)
    for i, (p, bs) in enumerate(zip(planes, bsz)):
        grid[i].reshape(43, B, B)[:, :bs, :bs] = p.reshape(43, bs, bs)

    def run(env, idx):
        monkeypatch.delenv("SAYURI_SE_SPLIT", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pipe = HipForwardPipe(g.weights_path, board_size=B, batch_size=128, fp16=True)
        try:
            ctx = pipe.ctx(0)
            outs = [hip_forward_raw(ctx, grid[i], [bsz[k] for k in i], B) for i in idx]
            rows = (_lib.KernelStat * 32)()
            k = _lib.hip().sayuri_hip_profile_run(ctx, rows, 32)
            names = {rows[i].name.decode(): rows[i].launches for i in range(k)}
            return outs, names
        finally:
            pipe.Destroy()

    alone = [[0], [1], [n - 2], [n - 1]]
    other = list(range(n - 1, -1, -2))  # every second sample, reversed: other mates, other places in the tiles
    (full, half, *solo), names = run({}, [list(range(n)), other] + alone)
    assert names.get("conv3x3_tower_sx", 0) == 13 and names.get("se_scale", 0) == 0, names  # the last batch profiled: one 9x9 alone
    (sep,), names0 = run({"SAYURI_SE_SPLIT": "0"}, [list(range(n))])
    assert names0.get("conv3x3_tower_sx", 0) == 0 and names0.get("se_scale", 0) == 13, names0
    scale = max(1.0, float(np.abs(sep[0]).max()))
    for a, b, what in zip(full, sep, ("prob", "pass", "misc", "own")):
        assert np.isfinite(a).all()
        assert float(np.abs(a - b).max()) <= FP16_ATOL * scale, (what, float(np.abs(a - b).max()))
    assert not all(np.array_equal(a, b) for a, b in zip(full, sep))  # the two forms round at different points
    for i in (0, n - 7, n - 6, n - 5, n - 2, n - 1):  # against the oracle: a drawn board, 5x5, 7x7, 19x19, 13x13, 9x9
        bs, s = bsz[i], bsz[i] ** 2
        e_prob, e_pass, e_misc, e_own = net.forward_raw(planes[i], bs)
        tol = FP16_ATOL * max(1.0, float(max(np.abs(e_prob).max(), np.abs(e_own).max(), np.abs(e_misc).max())))
        crop = lambda a: a.reshape(B, B)[:bs, :bs].ravel()
        for k in range(5):
            assert np.abs(crop(full[0][i, k]) - e_prob[k]).max() <= tol, (i, bs, "prob", k)
        assert np.abs(full[1][i] - e_pass).max() <= tol and np.abs(full[2][i] - e_misc).max() <= tol, (i, bs)
        assert np.abs(crop(full[3][i]) - e_own).max() <= tol, (i, bs, "own")
    for j, k in enumerate(other):  # the same bits among other batch mates ...
        for a, b in zip(full, half):
            assert np.array_equal(a[k], b[j]), ("mates", k, bsz[k])
    for idx, out in zip(alone, solo):  # ... and alone
        for a, b in zip(full, out):
            assert np.array_equal(a[idx[0]], b[0]), ("alone", idx[0], bsz[idx[0]])


def test_split_se_exchange_that_never_completes_fails_loudly(tmp_weights_dir, monkeypatch):
    """conv_board_sx.h: a workgroup waits for its siblings' partial sums inside the launch.  The wait is bounded: when a sibling never
    publishes (SAYURI_DEBUG_SX_STALL=1: channel tile 1 writes its granules under a wrong tag) every waiting workgroup gives up after
    ~0.3 s, sets a host-visible word, and the forward comes back as an ERROR -- not as a hang, and not as silently wrong numbers.  The
    context then goes on with the separate SE kernels; a healthy engine on the same device is unaffected."""
    import time
    from sayuri_amd.pipe import hip_forward_raw
    g = Golden("net_40b384", tmp_weights_dir)
    B, n = 19, 6
    planes = W.synthetic_planes(n, B, seed=6700)
    grid = np.ascontiguousarray(np.stack(planes), np.float32)
    monkeypatch.setenv("SAYURI_DEBUG_SX_STALL", "1")
    pipe = HipForwardPipe(g.weights_path, board_size=B, batch_size=8, fp16=True)
    monkeypatch.delenv("SAYURI_DEBUG_SX_STALL")
    try:
        t0 = time.time()
        with pytest.raises(RuntimeError, match="SE exchange"):
            hip_forward_raw(pipe.ctx(0), grid, [B] * n, B)
        assert time.time() - t0 < 120.0
        # the context goes on with the separate kernels (a wait that ran out once is not tried again): the next forward succeeds
        again = hip_forward_raw(pipe.ctx(0), grid, [B] * n, B)
    finally:
        pipe.Destroy()
    monkeypatch.setenv("SAYURI_SE_SPLIT", "0")
    pipe = HipForwardPipe(g.weights_path, board_size=B, batch_size=8, fp16=True)
    monkeypatch.delenv("SAYURI_SE_SPLIT")
    try:
        sep = hip_forward_raw(pipe.ctx(0), grid, [B] * n, B)
    finally:
        pipe.Destroy()
    for a, b in zip(again, sep):
        assert np.isfinite(a).all() and np.array_equal(a, b)  # ... with the bits of the separate kernels
    pipe = HipForwardPipe(g.weights_path, board_size=B, batch_size=8, fp16=True)
    try:
        out = hip_forward_raw(pipe.ctx(0), grid, [B] * n, B)
        assert all(np.isfinite(o).all() for o in out) and np.abs(out[0]).max() > 0
    finally:
        pipe.Destroy()


@pytest.mark.parametrize("fp16", [False, True], ids=["fp32", "fp16"])
def test_a_batch_of_many_tiny_boards_on_the_generic_kernel(fp16, tmp_weights_dir):
    """The reference accepts boards from 2x2 (types.h:23).  On a network whose channel count has no board kernel (6b x 96) the
    convolutions tile the batch's pixels across samples (conv_mfma.h); a 128-pixel tile over 2x2 boards touches 33 samples with a
    4x4 halo each.  Until round 6 the tile limits (24 samples, 352 halo positions) made the engine REFUSE a batch with two dozen
    2x2 / 3x3 boards in a row ("no conv tile configuration fits this batch geometry" -- found by the fuzz, seed 11); now such a
    batch is evaluated, and matches the oracle."""
    g = Golden("net_6b96", tmp_weights_dir)
    net = PortNet(g.weights_path)
    bsz = [2] * 40 + [3] * 30 + [9, 19, 2, 3, 5, 2]
    planes = W.synthetic_planes(len(bsz), bsz, seed=6800)
    pipe = HipForwardPipe(g.weights_path, board_size=19, batch_size=128, fp16=fp16)
    try:
        outs = pipe.BatchForward(planes, bsz)
    finally:
        pipe.Destroy()
    for i in (0, 17, 39, 40, 55, 69, 70, 71, 72, 73, 74, 75):
        exp = net.forward(planes[i], bsz[i])
        assert np.isfinite(outs[i]).all()
        assert np.abs(outs[i] - exp).max() <= (fp16_tol(exp) if fp16 else FP32_ATOL), (i, bsz[i])


def test_split_se_exchange_tags_wrap(tmp_weights_dir, monkeypatch):
    """conv_board_sx.h tags every exchanged value with the launch's epoch, which grows by one per SE layer launch; a slot's old tag
    must never equal a later epoch, so shortly before the 32-bit tags wrap the engine clears the buffer and starts over
    (Engine::forward).  A context whose tags start just below that point (SAYURI_DEBUG_SX_EPOCH0) crosses it within a few forwards:
    every forward, before and after, must carry the bits of an ordinary context."""
    from sayuri_amd.pipe import hip_forward_raw
    g = Golden("net_40b384", tmp_weights_dir)
    B = 19
    bsz = [19, 13, 9, 13, 19, 9, 9, 9, 13, 19]
    planes = W.synthetic_planes(len(bsz), bsz, seed=6900)
    grid = np.zeros((len(bsz), 43, B * B), np.float32)
    for i, (p, bs) in enumerate(zip(planes, bsz)):
        grid[i].reshape(43, B, B)[:, :bs, :bs] = p.reshape(43, bs, bs)
    pipe = HipForwardPipe(g.weights_path, board_size=B, batch_size=16, fp16=True)
    try:
        ref = hip_forward_raw(pipe.ctx(0), grid, bsz, B)
    finally:
        pipe.Destroy()
    monkeypatch.setenv("SAYURI_DEBUG_SX_EPOCH0", str(0xfff00000 - 30))  # 13 tags per forward: the third forward crosses the line
    pipe = HipForwardPipe(g.weights_path, board_size=B, batch_size=16, fp16=True)
    monkeypatch.delenv("SAYURI_DEBUG_SX_EPOCH0")
    try:
        for k in range(8):
            out = hip_forward_raw(pipe.ctx(0), grid, bsz, B)
            for a, b in zip(ref, out):
                assert np.array_equal(a, b), k
    finally:
        pipe.Destroy()


def test_chained_forward(tmp_weights_dir, monkeypatch):
    """configs[4]: a batch whose layers are more than one round of workgroups (40b x 384: 150 board tiles x 3 channel tiles on
    256 CUs) is run as chains of per-layer launches over groups of tiles, each on a stream of its own (Engine::forward; +6...10 %
    on the bench's configs[4] batch).  The chains cover disjoint tiles with the same kernels in the same buffers: the outputs must
    be the one-chain forward's BITS, for the engine's own choice (three chains here), for two and four, run after run, and through
    the queue.  (Round 5's first version failed exactly this -- a few dozen samples ~1e-4 off in two runs of three: the packed
    input's buffer, whose rows have the input convolution's channel stride, was recycled as a tower buffer, and a chain that ran
    ahead overwrote a later chain's input.  It now keeps a buffer of its own for the whole forward.)"""
    from sayuri_amd import _lib
    from sayuri_amd.pipe import hip_forward_raw
    g = Golden("net_40b384", tmp_weights_dir)
    rng = np.random.default_rng(56)
    B, n = 19, 256
    bsz = [int(b) for b in rng.choice([9, 13, 19], size=n)]
    planes = W.synthetic_planes(n, bsz, seed=5600 + n)
    grid = np.zeros((n, 43, B * B), np.float32)
    for i, (p, bs) in enumerate(zip(planes, bsz)):
        grid[i].reshape(43, B, B)[:, :bs, :bs] = p.reshape(43, bs, bs)
    switches = ("SAYURI_CHAINS", "SAYURI_CHAINS_SERIAL")

    def run(env, reps=2, queue=False):
        for k in switches:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pipe = HipForwardPipe(g.weights_path, board_size=B, batch_size=256, fp16=True)
        try:
            outs = [hip_forward_raw(pipe.ctx(0), grid, bsz, B) for _ in range(reps)]
            chains = _lib.hip().sayuri_hip_last_chains(pipe.ctx(0))
            q = pipe.BatchForward(planes, bsz) if queue else None
            return outs, chains, q
        finally:
            pipe.Destroy()

    ref, chains, q1 = run({"SAYURI_CHAINS": "1"}, queue=True)
    assert chains == 1 and all(np.array_equal(a, b) for a, b in zip(ref[0], ref[1]))
    assert np.abs(ref[0][0]).max() > 0
    for env, want in (({}, 3), ({"SAYURI_CHAINS": "2"}, 2), ({"SAYURI_CHAINS": "4"}, 4), ({"SAYURI_CHAINS": "3", "SAYURI_CHAINS_SERIAL": "1"}, 3)):
        outs, chains, q = run(env, reps=8, queue=not env)
        assert chains == want, (env, chains)
        for o in outs:
            for a, b, what in zip(ref[0], o, ("prob", "pass", "misc", "own")):
                assert np.array_equal(a, b), (env, what, float(np.abs(a - b).max()))
        if q is not None:   # the same batch through the pump (submit / wait)
            assert all(np.array_equal(x, y) for x, y in zip(q1, q))
    small = [int(b) for b in rng.choice([9, 13, 19], size=40)]   # 24 tiles x 3 = 72 workgroups: one round, nothing to fill
    pl = W.synthetic_planes(40, small, seed=77)
    gr = np.zeros((40, 43, B * B), np.float32)
    for i, (p, bs) in enumerate(zip(pl, small)):
        gr[i].reshape(43, B, B)[:, :bs, :bs] = p.reshape(43, bs, bs)
    for k in switches:
        monkeypatch.delenv(k, raising=False)
    pipe = HipForwardPipe(g.weights_path, board_size=B, batch_size=256, fp16=True)
    try:
        hip_forward_raw(pipe.ctx(0), gr, small, B)
        assert _lib.hip().sayuri_hip_last_chains(pipe.ctx(0)) == 1
    finally:
        pipe.Destroy()


def test_persistent_launch_with_more_tiles_than_cus(tmp_weights_dir):
    """The persistent tower launch has no grid-wide order between its workgroups' layers: a workgroup of the SECOND round starts
    when one of the first has walked the whole tower.  That is only sound if no workgroup ever writes bytes another one still has
    to read -- and rounds 3-4 broke it: the packed input's buffer (channel stride 64) was recycled as a tower buffer (stride
    256), so an early workgroup's third layer lay on top of a late workgroup's packed input.  Full-chip launches (256 tiles,
    all started together) won that race every time; 512 boards = 512 workgroups on 256 CUs (the batch size at which the
    full-width channel tile, and with it the persistent launch, is chosen again) lose it deterministically.  Every sample of the
    512-batch must be bit-equal to the same position in a batch of 256."""
    from sayuri_amd import _lib
    from sayuri_amd.pipe import hip_forward_raw
    g = Golden("net_20b256", tmp_weights_dir)
    B, n = 19, 512
    planes = W.synthetic_planes(n, B, seed=3200)
    grid = np.ascontiguousarray(np.stack(planes), np.float32)
    pipe = HipForwardPipe(g.weights_path, board_size=B, batch_size=n, fp16=True)
    try:
        ctx = pipe.ctx(0)
        big = hip_forward_raw(ctx, grid, [B] * n, B)
        # the persistent launch really ran (a launch per layer would pass this test whatever the buffers do)
        stat = _lib.KernelStat()
        lib = _lib.hip()
        ms = ctypes.c_float(0)
        lib.sayuri_hip_mark_kernel(ctx, b"tower_run")
        assert lib.sayuri_hip_time_runs(ctx, 1, ctypes.byref(ms)) == 0
        lib.sayuri_hip_timed_stat(ctx, ctypes.byref(stat))
        lib.sayuri_hip_mark_kernel(ctx, b"")
        assert stat.launches >= 1, "the 512-board batch did not take the persistent launch"
        assert all(np.isfinite(x).all() for x in big) and np.abs(big[0]).max() > 0
        for lo in (0, 256):
            part = hip_forward_raw(ctx, grid[lo:lo + 256], [B] * 256, B)
            for a, b, what in zip(big, part, ("prob", "pass", "misc", "own")):
                bad = [i for i in range(256) if not np.array_equal(a[lo + i], b[i])]
                assert not bad, (what, lo, len(bad), bad[:8])
    finally:
        pipe.Destroy()


def test_two_tickets_in_flight_give_the_solo_bits(tmp_weights_dir):
    """submit / wait with two batches in flight (what the pump does), a MIXED-size batch on the 20b x 256 network: its
    persistent launches have 150 workgroups, so the other ticket's kernels run beside them and workgroups start late.  Every
    batch must come back with the bits the same batch gives alone (round 4's engine: 3 batches of 100 off, the 141 samples of the
    late workgroups -- the recycled input buffer of the test above)."""
    import ctypes
    from sayuri_amd import _lib
    from sayuri_amd.pipe import hip_forward_raw
    g = Golden("net_20b256", tmp_weights_dir)
    rng = np.random.default_rng(56)
    B, n = 19, 256
    bsz = [int(b) for b in rng.choice([9, 13, 19], size=n)]
    planes = W.synthetic_planes(n, bsz, seed=5856)
    grid = np.zeros((n, 43, B * B), np.float32)
    for i, (p, bs) in enumerate(zip(planes, bsz)):
        grid[i].reshape(43, B, B)[:, :bs, :bs] = p.reshape(43, bs, bs)
    lib = _lib.hip()
    FP = ctypes.POINTER(ctypes.c_float)
    lib.sayuri_hip_host_alloc.restype = ctypes.c_void_p
    lib.sayuri_hip_host_alloc.argtypes = [ctypes.c_size_t]
    lib.sayuri_hip_host_free.argtypes = [ctypes.c_void_p]
    lib.sayuri_hip_submit.argtypes = [ctypes.c_void_p, ctypes.c_int, FP, ctypes.POINTER(ctypes.c_int), FP, FP, FP, FP, ctypes.POINTER(ctypes.c_int)]
    lib.sayuri_hip_wait.argtypes = [ctypes.c_void_p, ctypes.c_int]
    pipe = HipForwardPipe(g.weights_path, board_size=B, batch_size=n, fp16=True)
    raw = []
    try:
        ctx = pipe.ctx(0)
        ref = hip_forward_raw(ctx, grid, bsz, B)
        sizes = (grid.size, n * 5 * B * B, n * 5, n * 15, n * B * B)
        bufs = []
        for _ in range(2):
            ptrs = [lib.sayuri_hip_host_alloc(k * 4) for k in sizes]
            raw += ptrs
            np.ctypeslib.as_array(ctypes.cast(ptrs[0], FP), (grid.size,))[:] = grid.ravel()
            bufs.append(ptrs)
        hb = lib.sayuri_hip_host_alloc(n * 4)
        raw.append(hb)
        np.ctypeslib.as_array(ctypes.cast(hb, ctypes.POINTER(ctypes.c_int32)), (n,))[:] = np.asarray(bsz, np.int32)
        bp = ctypes.cast(hb, ctypes.POINTER(ctypes.c_int))
        tick = [ctypes.c_int(-1), ctypes.c_int(-1)]

        def submit(i):
            pl, pr, pa, mi, ow = bufs[i]
            assert lib.sayuri_hip_submit(ctx, n, ctypes.cast(pl, FP), bp, ctypes.cast(pr, FP), ctypes.cast(pa, FP), ctypes.cast(mi, FP),
                                         ctypes.cast(ow, FP), ctypes.byref(tick[i])) == 0, lib.sayuri_hip_last_error()

        def check(i, k):
            assert lib.sayuri_hip_wait(ctx, tick[i].value) == 0
            pl, pr, pa, mi, ow = bufs[i]
            got = (np.ctypeslib.as_array(ctypes.cast(pr, FP), (n, 5, B * B)), np.ctypeslib.as_array(ctypes.cast(pa, FP), (n, 5)),
                   np.ctypeslib.as_array(ctypes.cast(mi, FP), (n, 15)), np.ctypeslib.as_array(ctypes.cast(ow, FP), (n, B * B)))
            bad = sum(1 for s_ in range(n) if not all(np.array_equal(a[s_], b[s_]) for a, b in zip(ref, got)))
            assert bad == 0, (k, bad)

        submit(0); submit(1)
        for k in range(120):
            check(k & 1, k); submit(k & 1)
        check(0, 120); check(1, 121)
    finally:
        for q in raw:
            lib.sayuri_hip_host_free(ctypes.c_void_p(q))
        pipe.Destroy()


def test_packed_records_are_read_where_the_pump_has_them(tmp_weights_dir):
    """Round 5: the packed records of sayuri_hip_submit_packed are not copied when they lie in device-addressable pinned
    memory (sayuri_hip_host_alloc, the pump's buffers): pack_bits_kernel reads them across PCIe, so that batch k+1's first
    kernel does not queue behind batch k's downloads on the copy engine (profiles/r05_pump_gaps.txt).  Two tickets in flight
    with DIFFERENT records in their two buffers, refilled between rounds: every batch must come back with the bits the
    blocking, copying entry point (sayuri_hip_forward_packed) gives for the same records; records in pageable memory take the
    copy and give the same bits again."""
    from sayuri_amd import _lib
    from sayuri_amd.engine import pack_planes
    from sayuri_amd.pipe import hip_forward_packed_raw
    g = Golden("net_20b256", tmp_weights_dir)
    B, n, words = 19, 64, 37 * 12 + 8
    sets = []
    for k in range(3):
        bsz = [19] * n if k < 2 else [int(b) for b in np.random.default_rng(77).choice([9, 13, 19], size=n)]
        planes = W.synthetic_planes(n, bsz, seed=9100 + k)
        sets.append((np.stack([pack_planes(p, 37) for p in planes]).astype(np.uint32), bsz))
    lib = _lib.hip()
    FP = ctypes.POINTER(ctypes.c_float)
    IP = ctypes.POINTER(ctypes.c_int)
    lib.sayuri_hip_host_alloc.restype = ctypes.c_void_p
    lib.sayuri_hip_host_alloc.argtypes = [ctypes.c_size_t]
    lib.sayuri_hip_host_free.argtypes = [ctypes.c_void_p]
    lib.sayuri_hip_submit_packed.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, IP, FP, FP, FP, FP, IP]
    lib.sayuri_hip_wait.argtypes = [ctypes.c_void_p, ctypes.c_int]
    pipe = HipForwardPipe(g.weights_path, board_size=B, batch_size=n, fp16=True)
    raw = []
    try:
        ctx = pipe.ctx(0)
        refs = [hip_forward_packed_raw(ctx, rec, 37, bsz, B) for rec, bsz in sets]
        assert not np.array_equal(refs[0][0], refs[1][0])
        sizes = (n * words, n * 5 * B * B, n * 5, n * 15, n * B * B, n)
        bufs = []
        for _ in range(2):
            ptrs = [lib.sayuri_hip_host_alloc(k * 4) for k in sizes]
            assert all(ptrs)
            raw += ptrs
            bufs.append(ptrs)
        tick = [ctypes.c_int(-1), ctypes.c_int(-1)]
        holds = [None, None]

        def submit(i, k, pageable=False):
            rec, bsz = sets[k]
            pl, pr, pa, mi, ow, bz = bufs[i]
            np.ctypeslib.as_array(ctypes.cast(bz, ctypes.POINTER(ctypes.c_int32)), (n,))[:] = np.asarray(bsz, np.int32)
            if pageable:
                holds[i] = np.ascontiguousarray(rec)  # kept alive until the wait
                src = ctypes.c_void_p(holds[i].ctypes.data)
            else:
                np.ctypeslib.as_array(ctypes.cast(pl, ctypes.POINTER(ctypes.c_uint32)), (n * words,))[:] = rec.ravel()
                src = ctypes.c_void_p(pl)
            assert lib.sayuri_hip_submit_packed(ctx, n, src, 37, ctypes.cast(bz, IP), ctypes.cast(pr, FP), ctypes.cast(pa, FP),
                                                ctypes.cast(mi, FP), ctypes.cast(ow, FP), ctypes.byref(tick[i])) == 0, lib.sayuri_hip_last_error()

        def check(i, k):
            assert lib.sayuri_hip_wait(ctx, tick[i].value) == 0
            pl, pr, pa, mi, ow, bz = bufs[i]
            got = (np.ctypeslib.as_array(ctypes.cast(pr, FP), (n, 5, B * B)), np.ctypeslib.as_array(ctypes.cast(pa, FP), (n, 5)),
                   np.ctypeslib.as_array(ctypes.cast(mi, FP), (n, 15)), np.ctypeslib.as_array(ctypes.cast(ow, FP), (n, B * B)))
            for a, b, what in zip(refs[k], got, ("prob", "pass", "misc", "own")):
                assert np.array_equal(a, b), (k, what)

        order = [0, 1, 2, 1, 0, 2, 2, 0, 1, 0]
        submit(0, order[0]); submit(1, order[1])
        for j in range(2, len(order)):
            check(j & 1, order[j - 2]); submit(j & 1, order[j], pageable=(j >= 7))
        check(0, order[-2]); check(1, order[-1])
    finally:
        for q in raw:
            lib.sayuri_hip_host_free(ctypes.c_void_p(q))
        pipe.Destroy()


def test_fp16_error_is_measured_on_64_positions(tmp_weights_dir):
    """The fp16 gate is a measurement, not a guess: 64 positions of the 20b256 network against the reference's outputs
    (tests/golden/net_20b256_x64.npz).  Records max-abs error on the raw outputs and the reference's own SelfCheck L2
    (network.cc:333-359) to gpurun_out/fp16_error_20b256.json; the gate FP16_ATOL must stay >= 2x the measured maximum and
    the measured maximum must stay below it."""
    import json
    import os
    g = Golden("net_20b256_x64", tmp_weights_dir)
    cases = [(g.planes(c), c["board_size"], c["offset"], g.expected(c)) for c in g.cases]
    assert len(cases) == 64
    pipe = HipForwardPipe(g.weights_path, board_size=19, batch_size=64, fp16=True)
    try:
        outs = pipe.BatchForward([c[0] for c in cases], [c[1] for c in cases], offsets=[c[2] for c in cases])
    finally:
        pipe.Destroy()
    errs = [float(np.abs(o - c[3]).max()) for o, c in zip(outs, cases)]
    l2 = [self_check_l2(o, c[3], c[1]) for o, c in zip(outs, cases)]
    scale = float(max(np.abs(c[3]).max() for c in cases))
    gate = FP16_ATOL * max(1.0, scale)
    rec = {"positions": 64, "net": "20b256 seed 22", "max_abs": max(errs), "mean_abs_of_max": float(np.mean(errs)),
           "selfcheck_l2_max": max(l2), "selfcheck_l2_mean": float(np.mean(l2)), "gate": gate, "output_scale": scale}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "fp16_error_20b256.json"), "w") as f:
        json.dump(rec, f, indent=1)
    assert max(errs) <= gate and max(l2) <= 0.2
    assert gate >= 2 * max(errs), rec
