"""Fixture catalogue shared by tests/golden/make_golden.py (generator, dev container only)
and the tests that consume tests/golden/*.npz."""
from sayuri_amd import weights as W

B = W.BlockSpec


def spec_tiny_res():
    # 3 residual blocks (one with SE), 16 channels, heads 8
    return W.NetSpec(16, [B("ResidualBlock"), B("ResidualBlock", se=True), B("ResidualBlock")], 8, 8)


def spec_tiny_all():
    # every block family + RepLK policy head at 16 channels
    return W.NetSpec(16, [B("ResidualBlock", se=True), B("BottleneckBlock"), B("BottleneckBlock", se=True),
                          B("NestedBottleneckBlock"), B("NestedBottleneckBlock", se=True),
                          B("MixerBlock"), B("MixerBlock", se=True, kernel_size=5)],
                     8, 8, policy_head="RepLK")


def spec_tiny_relu_text():
    return W.NetSpec.residual(2, 16, 8, se_every=2, activation="relu")


def _acts(act):
    def f():
        return W.NetSpec(16, [B("ResidualBlock", se=True), B("MixerBlock")], 8, 8, activation=act)
    return f


_TENSORS_RES = ("input_conv.w", "input_conv.b", "input_conv.u", "tower.1.conv2.w", "tower.1.conv2.u",
                "tower.1.squeeze.w", "tower.1.excite.b", "p_hd_conv.w", "p_inter_fc.w", "prob_conv.w",
                "pass_fc.b", "v_hd_conv.b", "v_inter_fc.w", "v_ownership.w", "v_misc.w")

_CASES_SMALL = ((9, 0, 101), (13, 1, 102), (19, 0, 103), (19, 4, 104), (7, 2, 105))

FIXTURES = [
    dict(name="tiny_res", spec=spec_tiny_res, seed=11, commit_weights=True, winograd=(1, 0),
         tensors=_TENSORS_RES, cases=_CASES_SMALL),
    dict(name="tiny_all", spec=spec_tiny_all, seed=12, commit_weights=True, winograd=(1, 0),
         tensors=("tower.1.pre_btl_conv.w", "tower.3.conv4.u", "tower.5.dw_conv.w", "tower.5.dw_conv.b",
                  "tower.6.conv1.w", "p_dw_conv.w", "p_pt_conv.b"),
         cases=_CASES_SMALL),
    dict(name="tiny_relu_text", spec=spec_tiny_relu_text, seed=13, commit_weights=True, binary=False,
         winograd=(1,), tensors=("input_conv.w", "v_misc.b"), cases=((9, 0, 201), (19, 3, 202))),
] + [
    dict(name=f"tiny_act_{a}", spec=_acts(a), seed=14, commit_weights=False, winograd=(1,),
         cases=((9, 0, 301), (19, 1, 302)))
    for a in ("identity", "elu", "selu", "gelu", "swish", "hardswish")
] + [
    # BASELINE.json configs[0]: 9x9, 6-block x 96-filter net (weights regenerated from the seed)
    dict(name="net_6b96", spec=W.spec_6b96, seed=21, commit_weights=False, winograd=(1,),
         tensors=("tower.2.conv1.b",), cases=((9, 0, 401), (9, 2, 402), (19, 0, 403))),
    # BASELINE.json configs[1..3] network: 19x19, 20-block x 256-filter
    dict(name="net_20b256", spec=W.spec_20b256, seed=22, commit_weights=False, winograd=(1,),
         cases=((19, 0, 501), (19, 0, 502), (13, 1, 503))),
    # the same network on 64 more 19x19 positions: the sample the fp16 engine's error is measured on (planes are
    # regenerated from their seeds, only the reference's outputs are stored)
    dict(name="net_20b256_x64", spec=W.spec_20b256, seed=22, commit_weights=False, winograd=(1,), store_planes=False,
         cases=tuple((19, i % 5, 700 + i) for i in range(64))),
    # fp16 stress: the 20b x 256 architecture with BN statistics that let the residual stream grow to |x| ~ 1.4e3 by block 20
    # (sayuri_amd.weights.spec_20b256_hot; the ordinary random-init nets keep it at O(1), so the fp16 gate was calibrated on
    # small activations only).  fp16 storage of 40 accumulated residual additions at that magnitude is what is checked.
    dict(name="net_20b256_hot", spec=lambda: W.spec_20b256_hot(5.5), seed=24, commit_weights=False, winograd=(1,), store_planes=False,
         cases=tuple((19, i % 5, 800 + i) for i in range(8))),
    # BASELINE.json configs[4] network: 40-block x 384-filter, boards 19 / 13 / 9
    dict(name="net_40b384", spec=W.spec_40b384, seed=23, commit_weights=False, winograd=(1,), store_planes=False,
         cases=((19, 0, 601), (13, 1, 602), (9, 2, 603), (19, 3, 604))),
]
