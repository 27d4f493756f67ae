import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import mpn_oracle
    mpn_oracle.build()
    return mpn_oracle


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import multipathnet_amd
    multipathnet_amd.load()
    return torch.device("cuda", 0)


def random_scored_boxes(rng, n, regime="distinct", span=1000.0, lo=16.0, hi=400.0):
    """SURVEY §8d NMS micro-inputs: distinct random / heavy ties (1/64 quantised) / saturated (many exactly 1.0f)."""
    c = rng.uniform(0, span, (n, 2))
    wh = np.exp(rng.uniform(np.log(lo), np.log(hi), (n, 2)))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1)
    s = rng.uniform(0, 1, (n, 1))
    if regime == "ties":
        s = np.round(s * 64) / 64
    elif regime == "saturated":
        s = np.where(s > 0.5, 1.0, s)
    elif regime == "allequal":
        s = np.full_like(s, 0.5)
    return np.concatenate([b, s], 1).astype(np.float32)
