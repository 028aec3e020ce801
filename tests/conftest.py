import importlib.util
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: BASELINE-size end-to-end parity (CPU oracle at batch 32: minutes)")


def _load_gen():
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(GOLDEN, "gen_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def gen():
    """tests/golden/gen_golden.py as a module (input builders for the golden cases)."""
    return _load_gen()


@pytest.fixture(scope="session")
def decode_golden():
    z = np.load(os.path.join(GOLDEN, "decode_golden.npz"))
    with open(os.path.join(GOLDEN, "decode_golden.json")) as f:
        meta = json.load(f)
    return z, meta


@pytest.fixture(scope="session")
def net_golden():
    z = np.load(os.path.join(GOLDEN, "net_golden.npz"))
    with open(os.path.join(GOLDEN, "net_golden.json")) as f:
        meta = json.load(f)
    return z, meta


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
