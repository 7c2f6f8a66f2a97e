import os
import sys

import numpy as np
import pytest

# torch carries its own HIP runtime: it has to be loaded before libmdisp.so pulls in the system one,
# otherwise torch.cuda finds no devices in the tests that use torch for device buffers.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.build()
    return o


@pytest.fixture(scope="session")
def api():
    from cameracalibration_amd import api as a
    return a


def rand_pair(seed, H, W, D):
    """Small synthetic rectified pair without touching the GPU library."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(H, W + D), dtype=np.uint8)
    # light horizontal smoothing so neighbouring pixels correlate
    b = base.astype(np.int32)
    sm = (b + np.roll(b, 1, 1) + np.roll(b, -1, 1) + np.roll(b, 1, 0)) // 4
    right = sm[:, :W].astype(np.uint8)
    shift = max(1, D // 3)
    left = np.roll(sm, shift, axis=1)[:, :W].astype(np.uint8)
    noise = rng.integers(-2, 3, size=(H, W))
    left = np.clip(left.astype(np.int32) + noise, 0, 255).astype(np.uint8)
    return left, right
