"""CPU-only checks of the product host side: the weights loader against the oracle and the
reference-generated goldens, the C-ABI export list, and error behaviour without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from _golden import Golden
from _oracle import PortNet
from golden_specs import FIXTURES
from sayuri_amd import _build, _lib
from sayuri_amd.pipe import Weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ALL_TENSORS = ["input_conv", "p_hd_conv", "p_inter_fc", "prob_conv", "pass_fc", "v_hd_conv", "v_inter_fc",
               "v_ownership", "v_misc"]


@pytest.fixture(scope="module", autouse=True)
def built():
    _build.build_host()


def _block_layers(binfo):
    t, se = binfo[0], binfo[1]
    names = {1: ["conv1", "conv2"], 2: ["pre_btl_conv", "conv1", "conv2", "post_btl_conv"],
             3: ["pre_btl_conv", "conv1", "conv2", "conv3", "conv4", "post_btl_conv"],
             4: ["dw_conv", "conv1", "conv2"]}[t]
    return names + (["squeeze", "excite"] if se else [])


@pytest.mark.parametrize("name", [fx["name"] for fx in FIXTURES if fx["name"].startswith("tiny")] + ["net_6b96"])
def test_product_loader_matches_oracle_bit_for_bit(name, tmp_weights_dir):
    """Same file -> same folded tensors, exactly (both are plain IEEE fp32, no fast-math)."""
    g = Golden(name, tmp_weights_dir)
    w = Weights(g.weights_path)
    o = PortNet(g.weights_path)
    assert w.info == o.info == list(g.data["info"])
    names = list(ALL_TENSORS)
    if w.info[11]:
        names += ["p_dw_conv", "p_pt_conv"]
    for i in range(w.info[2]):
        assert w.block_info(i) == o.block_info(i) == list(g.data["blocks"][i])
        names += [f"tower.{i}.{l}" for l in _block_layers(w.block_info(i))]
    for n in names:
        for kind in ("w", "b"):
            a, b = w.tensor(f"{n}.{kind}"), o.tensor(f"{n}.{kind}")
            assert a is not None and b is not None, n
            np.testing.assert_array_equal(a, b, err_msg=f"{n}.{kind}")
    # and against the reference loader's tensors stored in the golden (fast-math build: ~1 ulp)
    for tn, exp in g.tensors().items():
        if tn.endswith(".u"):
            continue  # this backend keeps no Winograd-transformed copy
        np.testing.assert_allclose(w.tensor(tn), exp, rtol=2e-6, atol=1e-7, err_msg=tn)


def test_loader_error_behaviour(tmp_path):
    # reference: load failures leave weights->loaded false and log the cause (loader.cc:61-64)
    with pytest.raises(RuntimeError, match="Couldn't open"):
        Weights(str(tmp_path / "nope.bin"))
    p = tmp_path / "bad.txt"
    p.write_text("get nothing\n")
    with pytest.raises(RuntimeError, match="not acceptable"):
        Weights(str(p))
    g = Golden("tiny_res", str(tmp_path))
    blob = open(g.weights_path, "rb").read()
    q = tmp_path / "trunc.bin"
    q.write_bytes(blob[:len(blob) // 3])
    with pytest.raises(RuntimeError):
        Weights(str(q))
    v6 = tmp_path / "v6.bin"
    v6.write_bytes(blob.replace(b"Version 5", b"Version 6"))
    with pytest.raises(RuntimeError, match="do not support this version"):
        Weights(str(v6))


def test_c_abi_exports_every_declared_symbol():
    """include/sayuri_hip.h is the boundary: every function it declares must be exported."""
    header = open(os.path.join(ROOT, "include", "sayuri_hip.h")).read()
    declared = set(re.findall(r"\b(sayuri_hip_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.HIP_SYMBOLS), declared ^ set(_lib.HIP_SYMBOLS)
    lib = ctypes.CDLL(_build.HIP_SO)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_host_library_exports_every_declared_symbol():
    """include/sayuri_engine.h lists the host library's C entry points: declared == exported."""
    import subprocess
    header = open(os.path.join(ROOT, "include", "sayuri_engine.h")).read()
    declared = set(re.findall(r"\b(sayuri_[a-z_]+)\s*\(", header))
    nm = subprocess.run(["nm", "-D", "--defined-only", _build.HOST_SO], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in nm.splitlines() if " T sayuri_" in line}
    assert declared == exported, declared ^ exported


@pytest.mark.skipif(_lib.hip().sayuri_hip_device_count() > 0, reason="a GPU is present")
def test_fails_loudly_without_gpu(tmp_weights_dir):
    from sayuri_amd.pipe import HipForwardPipe
    g = Golden("tiny_res", tmp_weights_dir)
    with pytest.raises(RuntimeError, match="No executable GPU device"):
        HipForwardPipe(g.weights_path)


@pytest.mark.parametrize("fibers,threads,rounds", [(1, 1, 5), (64, 3, 50), (1000, 8, 20), (4096, 4, 5)])
def test_fiber_runtime_selftest(fibers, threads, rounds):
    """csrc/host/fiber.h, the M:N scheduling of self-play games: every fiber is resumed exactly when its word changes,
    keeps its stack across switches and finishes; thousands of fibers on a handful of OS threads."""
    import ctypes
    h = _lib.host()
    h.sayuri_fiber_selftest.restype = ctypes.c_long
    assert h.sayuri_fiber_selftest(fibers, threads, rounds) == fibers * rounds
