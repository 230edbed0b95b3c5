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


def synth(kind: str, n: int, seed: int) -> np.ndarray:
    """Deterministic synthetic waveforms built from a 64-bit integer PRNG (bit-identical everywhere)."""
    rng = np.random.default_rng(seed)
    u = rng.integers(-(1 << 23), 1 << 23, size=n, dtype=np.int64).astype(np.float64) / float(1 << 23)   # uniform [-1, 1)
    if kind == "uniform":
        x = u
    elif kind == "noise0.1":       # sum of 4 uniforms ~ bell shaped, std ~0.1
        v = rng.integers(-(1 << 23), 1 << 23, size=(3, n), dtype=np.int64).astype(np.float64) / float(1 << 23)
        x = 0.0866 * (u + v.sum(axis=0))
    elif kind == "sine+noise":
        t = np.arange(n, dtype=np.float64)
        x = 0.5 * np.sin(2 * np.pi * 440.0 * t / 16000.0) + 1e-3 * u
    elif kind == "silence":
        x = np.zeros(n)
    elif kind == "impulse":
        x = np.zeros(n)
        x[n // 3] = 1.0
    elif kind == "dc":
        x = 0.25 + 0.01 * u
    elif kind == "quiet":
        x = 1e-3 * u
    else:
        raise ValueError(kind)
    return np.clip(x, -1.0, 1.0).astype(np.float32)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
