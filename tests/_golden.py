"""Helpers to read tests/golden/*.npz (see tests/golden/make_golden.py)."""
import hashlib
import json
import os

import numpy as np

from golden_specs import FIXTURES
from sayuri_amd import weights as W

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BY_NAME = {fx["name"]: fx for fx in FIXTURES}


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


class Golden:
    def __init__(self, name, workdir):
        self.fx = BY_NAME[name]
        self.name = name
        self.data = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
        self.cases = json.loads(str(self.data["cases"]))
        committed = os.path.join(GOLDEN_DIR, f"{name}.bin")
        if self.fx.get("commit_weights", False):
            self.weights_path = committed
        else:
            self.weights_path = os.path.join(workdir, f"{name}.bin")
            if not os.path.exists(self.weights_path):
                W.write_weights(self.weights_path, self.fx["spec"](), seed=self.fx["seed"],
                                binary=self.fx.get("binary", True))
        # the golden outputs belong to exactly this weight file
        assert sha256(self.weights_path) == str(self.data["sha256"]), "weight generator drifted"

    def planes(self, case):
        p = self.data[f"planes:{case['key']}"]
        if p.size == 0:  # store_planes=False fixtures: the generator's seeded planes, regenerated
            p = W.synthetic_planes(1, case["board_size"], seed=case["planes_seed"])[0]
        return p

    def expected(self, case):
        return self.data[f"out:{case['key']}"]

    def tensors(self):
        return {k[len("tensor:"):]: self.data[k] for k in self.data.files if k.startswith("tensor:")}
