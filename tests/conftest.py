import contextlib
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


HOOK_DEFAULTS = dict(conv_variant=0, conv_split=0, gemm_split=0, fuse_pool=1, nms_force_exact=0, fp32_pf=1, bf16_dma=1, bf16_dma_tn=0,
                     bf16_fast_pool=3, graph_fuse=3, first_k36=1, wino_tc=0, roi_pool_pm=1)


@contextlib.contextmanager
def hooks(**settings):
    """Forced kernel variants / split factors for a test: `with hooks(conv_variant=7, conv_split=2): ...`.

    The PRODUCT library (libmpn_hip.so) has no such switches — its knobs are compile-time constants — so a non-default
    setting runs the block on libmpn_hip_dbg.so (same sources, -DMPN_DEBUG_HOOKS; multipathnet_amd._lib.debug_hooks) with
    mpn_debug_set_<name>(value) applied and reset afterwards.  All-default settings leave the block on the product library:
    the default dispatch is always tested on the library that ships."""
    from multipathnet_amd import _lib
    changed = {k: v for k, v in settings.items() if HOOK_DEFAULTS[k] != v}
    if not changed:
        yield None
        return
    with _lib.debug_hooks() as lib:
        for k, v in changed.items():
            getattr(lib, "mpn_debug_set_" + k)(v)
        try:
            yield lib
        finally:
            for k in changed:
                getattr(lib, "mpn_debug_set_" + k)(HOOK_DEFAULTS[k])


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
