import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """One tests/golden/*.npz fixture: .meta (dict), .sd / .ins / .outs (dict[str, Tensor])."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.sd, self.ins, self.outs = {}, {}, {}
        for k in z.files:
            if k == "meta":
                continue
            kind, key = k.split("/", 1)
            {"sd": self.sd, "in": self.ins, "out": self.outs}[kind][key] = torch.from_numpy(z[k])


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return load


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b| -- the 'relative' of the 1e-4 fp32 parity bar (north_star)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def report(line: str) -> None:
    """Numbers a passing test wants on record (ids that differ, near-tie counts, streaming work ratio): printed, and appended to
    gpurun_out/parity_report.txt so that they come back from the GPU box (copied into profiles/ for the round)."""
    print(line)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_report.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
