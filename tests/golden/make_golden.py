#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE ITSELF.

Run in the dev container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py [fixture names ...]

For each fixture it writes a synthetic network file with sayuri_amd.weights (seeded
numpy), loads it with the reference's own DNNLoader and evaluates seeded planes with
the reference's own BlasForwardPipe::Forward (oracle/_ref/libsayuri_ref.so, compiled
from the unmodified reference sources by oracle/Makefile).  Stored per fixture:

  * <name>.npz   -- inputs (planes), expected raw outputs, a few post-fold tensors
                    (folded conv weights/biases, Winograd U), net info and the sha256
                    of the generated weight file
  * <name>.bin   -- the weight file itself, for the small nets only

Big nets (6b96, 20b256) are re-generated from (spec, seed) by the tests; the sha256
guards against generator drift.  The reference ships no golden vectors of its own
(SURVEY.md section 4), so these files are the pin of the oracle.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from sayuri_amd import weights as W  # noqa: E402
from _oracle import RefNet  # noqa: E402
from golden_specs import FIXTURES  # noqa: E402


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def main():
    only = set(sys.argv[1:])  # optional: fixture names to (re)generate; default = all
    for fx in FIXTURES:
        name = fx["name"]
        if only and name not in only:
            continue
        spec = fx["spec"]()
        keep = fx.get("commit_weights", False)
        wpath = os.path.join(HERE, f"{name}.bin") if keep else f"/tmp/golden_{name}.bin"
        W.write_weights(wpath, spec, seed=fx["seed"], binary=fx.get("binary", True))
        out = {"sha256": np.array(sha256(wpath))}
        cases = []
        for wino in fx.get("winograd", (1,)):
            net = RefNet(wpath, bool(wino))
            out["info"] = np.array(net.info, np.int32)
            out["blocks"] = np.array([net.block_info(i) for i in range(net.info[2])], np.int32).reshape(-1, 5)
            for tn in fx.get("tensors", ()):
                t = net.tensor(tn)
                assert t is not None, tn
                out[f"tensor:{tn}"] = t
            for ci, (bs, offset, pseed) in enumerate(fx["cases"]):
                planes = W.synthetic_planes(1, bs, seed=pseed)[0]
                res = net.forward(planes, bs, offset=offset)
                key = f"w{wino}_c{ci}"
                out[f"planes:{key}"] = planes if fx.get("store_planes", True) else np.zeros(0, np.float32)
                out[f"out:{key}"] = res
                cases.append({"key": key, "winograd": int(wino), "board_size": bs, "offset": offset,
                              "planes_seed": pseed})
        out["cases"] = np.array(json.dumps(cases))
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(f"{name}: {len(cases)} cases, weights sha256 {str(out['sha256'])[:12]}"
              f"{' (committed)' if keep else ''}")


if __name__ == "__main__":
    main()
