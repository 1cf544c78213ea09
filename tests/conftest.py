"""Shared test plumbing.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol export (no GPU needed).
`-m gpu`      : parity tests proper - they call the HIP kernels through the C-ABI on a real MI355X.
"""
import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd")
GOLDEN_DIR = os.path.join(REPO, "tests", "golden")
for p in (PKG_DIR, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    arrays = np.load(os.path.join(GOLDEN_DIR, "qim_dct_golden.npz"), allow_pickle=False)
    with open(os.path.join(GOLDEN_DIR, "qim_dct_golden.json")) as fh:
        meta = json.load(fh)
    return arrays, meta


def gpu_present() -> bool:
    """True when a HIP device is visible (checked without importing torch)."""
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)
