"""Synthetic Sayuri network files (the on-disk contract of the hot path).

The engine reads the reference's weight-file format (reference
src/neural/loader.cc:67-121 `DNNLoader::Parse`, written by
train/torch/network.py:1399-1439 `transfer_to_bin`).  There is no network access
for real checkpoints, so tests and bench.py generate random-init networks of a
named architecture with this module and write them in that format:

    get main
    get info ... end info          (key value lines)
    get stack ... end stack        (one block name per line, e.g. ResidualBlock-SE)
    get struct ... end struct      (Convolution i o k / DepthwiseConvolution i o k /
                                    BatchNorm c / FullyConnect i o)
    get parameters
      <tensor stream>              float32bin: little-endian f32 values ended by
                                   the word 0xFFFFFFFF; text: one line per tensor
    end parameters
    end main

Per conv+BN pair the file holds: conv W [K][C][k][k], conv bias [K] (zeros, as the
trainer writes them), BN mean [K], BN *stddev* [K] (Version>=2; loader.cc:914-925).
The tensor order is the one `DNNLoader::FillWeights`/`FillBlock` consume
(loader.cc:358-773).

Only numpy is used, so the same seed gives byte-identical files here and on the
GPU box.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

INPUT_CHANNELS = 43  # v3+ encoder (loader.cc:225-231)
POLICY_OUTS = 5
VALUE_MISC = 15
END_WORD = b"\xff\xff\xff\xff"

_GAIN = {  # train/torch/network.py:38-57
    "identity": 1.0,
    "relu": math.sqrt(2.0),
    "elu": math.sqrt(1.55052),
    "selu": 0.75,
    "gelu": math.sqrt(2.351718),
    "mish": math.sqrt(2.210277),
    "swish": math.sqrt(2.0),
    "hardswish": math.sqrt(2.0),
}


@dataclass
class BlockSpec:
    kind: str = "ResidualBlock"  # ResidualBlock | BottleneckBlock | NestedBottleneckBlock | MixerBlock
    se: bool = False
    bottleneck_channels: Optional[int] = None  # default channels // 2
    ffn_channels: Optional[int] = None  # default int(1.5 * channels)
    kernel_size: int = 7  # mixer depthwise kernel

    @property
    def name(self) -> str:
        return self.kind + ("-SE" if self.se else "")


@dataclass
class NetSpec:
    channels: int
    blocks: List[BlockSpec]
    policy_channels: int
    value_channels: int
    activation: str = "mish"
    se_ratio: int = 4
    policy_head: str = "Normal"  # Normal | RepLK
    replk_kernel: int = 7
    version: int = 5
    nntype: str = "Residual"
    # generator-only knob (not part of the file format): multiplies the scale of every branch that is added back to the
    # skip.  1.0 keeps the residual stream O(1) through deep towers; larger values make the stream GROW from block to
    # block (the fp16 stress fixture, tests/golden_specs.py: |x| ~ 1e3 at the end of a 20-block tower)
    branch_scale: float = 1.0

    @staticmethod
    def residual(nblocks: int, channels: int, head_channels: int, se_every: int = 3,
                 activation: str = "mish") -> "NetSpec":
        """The shape of bash/configs/selfplay-setting.json: every `se_every`-th block has SE."""
        blocks = [BlockSpec("ResidualBlock", se=(se_every > 0 and (i + 1) % se_every == 0))
                  for i in range(nblocks)]
        return NetSpec(channels, blocks, head_channels, head_channels, activation=activation)


# named architectures used by BASELINE.json's configs
def spec_6b96() -> NetSpec:
    return NetSpec.residual(6, 96, 24)


def spec_20b256() -> NetSpec:
    return NetSpec.residual(20, 256, 32)


def spec_40b384() -> NetSpec:
    return NetSpec.residual(40, 384, 48)


def spec_20b256_hot(branch_scale: float = 3.0) -> NetSpec:
    """configs[1]'s architecture with BN statistics that let the residual stream grow: the fp16 stress network."""
    s = NetSpec.residual(20, 256, 32)
    s.branch_scale = branch_scale
    return s


Layer = Tuple[str, Tuple[int, ...], List[np.ndarray]]  # (struct line, shape ints, tensors)


class _Gen:
    def __init__(self, spec: NetSpec, seed: int):
        self.spec = spec
        self.rng = np.random.default_rng(seed)
        self.layers: List[Tuple[str, List[np.ndarray]]] = []
        self.named: Dict[str, np.ndarray] = {}

    def _xavier(self, shape, fan_in, fan_out, act) -> np.ndarray:
        std = _GAIN[act] * math.sqrt(2.0 / (fan_in + fan_out))
        return (self.rng.standard_normal(shape) * std).astype(np.float32)

    def _bn(self, c: int, out_scale: float) -> Tuple[np.ndarray, np.ndarray]:
        mean = (self.rng.standard_normal(c) * 0.1).astype(np.float32)
        var = self.rng.uniform(0.5, 1.5, c)
        std = (np.sqrt(var) / out_scale).astype(np.float32)  # file stores stddev (v2+)
        return mean, std

    def conv_block(self, name, cin, cout, k, act, out_scale=1.0):
        w = self._xavier((cout, cin, k, k), cin * k * k, cout * k * k, act)
        b = np.zeros(cout, np.float32)
        mean, std = self._bn(cout, out_scale)
        self.layers.append((f"Convolution {cin} {cout} {k}", [w, b]))
        self.layers.append((f"BatchNorm {cout}", [mean, std]))
        self.named.update({f"{name}.w": w, f"{name}.b": b, f"{name}.bn_mean": mean,
                           f"{name}.bn_std": std})

    def dwconv_block(self, name, c, k, act, out_scale=1.0):
        w = self._xavier((c, 1, k, k), k * k, k * k, act)
        b = (self.rng.standard_normal(c) * 0.05).astype(np.float32)
        mean, std = self._bn(c, out_scale)
        self.layers.append((f"DepthwiseConvolution 1 {c} {k}", [w, b]))
        self.layers.append((f"BatchNorm {c}", [mean, std]))
        self.named.update({f"{name}.w": w, f"{name}.b": b, f"{name}.bn_mean": mean,
                           f"{name}.bn_std": std})

    def conv(self, name, cin, cout, k, act):
        w = self._xavier((cout, cin, k, k), cin * k * k, cout * k * k, act)
        b = (self.rng.standard_normal(cout) * 0.05).astype(np.float32)
        self.layers.append((f"Convolution {cin} {cout} {k}", [w, b]))
        self.named.update({f"{name}.w": w, f"{name}.b": b})

    def fc(self, name, cin, cout, act):
        w = self._xavier((cout, cin), cin, cout, act)
        b = (self.rng.standard_normal(cout) * 0.05).astype(np.float32)
        self.layers.append((f"FullyConnect {cin} {cout}", [w, b]))
        self.named.update({f"{name}.w": w, f"{name}.b": b})

    def build(self):
        s = self.spec
        act = s.activation
        c = s.channels
        nb = max(len(s.blocks), 1)
        # keep the residual stream O(1) through deep towers: the branch that is
        # added back to the skip is scaled like a trained final-BN gamma
        res_scale = s.branch_scale / math.sqrt(nb)
        self.conv_block("input_conv", INPUT_CHANNELS, c, 3, act)
        for i, blk in enumerate(s.blocks):
            p = f"tower.{i}"
            if blk.kind == "ResidualBlock":
                self.conv_block(f"{p}.conv1", c, c, 3, act)
                self.conv_block(f"{p}.conv2", c, c, 3, "identity", res_scale)
            elif blk.kind == "BottleneckBlock":
                inner = blk.bottleneck_channels or c // 2
                self.conv_block(f"{p}.pre_btl_conv", c, inner, 1, act)
                self.conv_block(f"{p}.conv1", inner, inner, 3, act)
                self.conv_block(f"{p}.conv2", inner, inner, 3, act)
                self.conv_block(f"{p}.post_btl_conv", inner, c, 1, "identity", res_scale)
            elif blk.kind == "NestedBottleneckBlock":
                inner = blk.bottleneck_channels or c // 2
                self.conv_block(f"{p}.pre_btl_conv", c, inner, 1, act)
                self.conv_block(f"{p}.conv1", inner, inner, 3, act)
                self.conv_block(f"{p}.conv2", inner, inner, 3, "identity", 0.7)
                self.conv_block(f"{p}.conv3", inner, inner, 3, act)
                self.conv_block(f"{p}.conv4", inner, inner, 3, "identity", 0.7)
                self.conv_block(f"{p}.post_btl_conv", inner, c, 1, "identity", res_scale)
            elif blk.kind == "MixerBlock":
                ffn = blk.ffn_channels or int(1.5 * c)
                self.dwconv_block(f"{p}.dw_conv", c, blk.kernel_size, act, 0.5)
                self.conv_block(f"{p}.conv1", c, ffn, 1, act)
                self.conv_block(f"{p}.conv2", ffn, c, 1, "identity", res_scale)
            else:
                raise ValueError(f"unknown block kind {blk.kind}")
            if blk.se:
                se = c // s.se_ratio
                self.fc(f"{p}.squeeze", 3 * c, se, act)
                self.fc(f"{p}.excite", se, 2 * c, "identity")
        pc, vc = s.policy_channels, s.value_channels
        self.conv_block("p_hd_conv", c, pc, 1, act)
        if s.policy_head == "RepLK":
            self.dwconv_block("p_dw_conv", pc, max(s.replk_kernel, 7), act)
            self.conv_block("p_pt_conv", pc, pc, 1, act)
        self.fc("p_inter_fc", 3 * pc, pc, act)
        self.conv("prob_conv", pc, POLICY_OUTS, 1, "identity")
        self.fc("pass_fc", pc, POLICY_OUTS, "identity")
        self.conv_block("v_hd_conv", c, vc, 1, act)
        self.fc("v_inter_fc", 3 * vc, 3 * vc, act)
        self.conv("v_ownership", vc, 1, 1, "identity")
        self.fc("v_misc", 3 * vc, VALUE_MISC, "identity")
        return self


def generate(spec: NetSpec, seed: int = 0):
    """Return (layers, named) for `spec`: the file-order layer list and a dict of the raw
    (un-folded) tensors keyed like the engine's layer names."""
    g = _Gen(spec, seed).build()
    return g.layers, g.named


def _header(spec: NetSpec, float_type: Optional[str]) -> str:
    lines = ["get main", "get info", f"NNType {spec.nntype}", f"Version {spec.version}"]
    if float_type:
        lines.append(f"FloatType {float_type}")
    lines += [f"InputChannels {INPUT_CHANNELS}", f"ResidualChannels {spec.channels}",
              f"ResidualBlocks {len(spec.blocks)}",
              f"PolicyHeadChannels {spec.policy_channels}",
              f"ValueHeadChannels {spec.value_channels}", f"ValueMisc {VALUE_MISC}",
              f"PolicyHeadType {spec.policy_head}", f"ActivationFunction {spec.activation}",
              "end info", "get stack"]
    lines += [b.name for b in spec.blocks]
    lines += ["end stack"]
    return "\n".join(lines) + "\n"


def write_weights(path: str, spec: NetSpec, seed: int = 0, binary: bool = True) -> Dict[str, np.ndarray]:
    """Write a random-init network of architecture `spec` to `path` in the reference's
    file format (float32bin when `binary`, else the text form).  Returns the raw tensors."""
    layers, named = generate(spec, seed)
    tmp = f"{path}.tmp.{os.getpid()}"
    with open(tmp, "wb") as f:
        f.write(_header(spec, "float32bin" if binary else None).encode())
        f.write(b"get struct\n")
        for line, _ in layers:
            f.write((line + "\n").encode())
        f.write(b"end struct\nget parameters\n")
        for _, tensors in layers:
            for t in tensors:
                flat = np.ascontiguousarray(t, dtype="<f4").ravel()
                if binary:
                    f.write(flat.tobytes())
                    f.write(END_WORD)
                else:
                    f.write((" ".join(repr(float(v)) for v in flat) + "\n").encode())
        f.write(b"end parameters\nend main")
    os.replace(tmp, path)
    return named


def synthetic_planes(n: int, board_sizes, seed: int = 0, komi: float = 7.5) -> List[np.ndarray]:
    """Seeded synthetic encoder planes: per sample a [43][bs*bs] float32
    array packed with the sample's own board stride (InputData, network_basic.h:23-34).
    Channels 0..36 Bernoulli(0.2); 37 rule = 0; 38 wave = 0; 39/40 = +-komi/20;
    41 = bs*bs/361; 42 = 1 (the on-board mask plane)."""
    rng = np.random.default_rng(seed)
    if np.isscalar(board_sizes):
        board_sizes = [int(board_sizes)] * n
    out = []
    for i in range(n):
        bs = int(board_sizes[i])
        s = bs * bs
        p = np.zeros((INPUT_CHANNELS, s), np.float32)
        p[:37] = (rng.random((37, s)) < 0.2).astype(np.float32)
        p[39] = komi / 20.0
        p[40] = -komi / 20.0
        p[41] = s / 361.0
        p[42] = 1.0
        out.append(p)
    return out
