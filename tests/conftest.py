import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests hand torch device tensors to libmsiren: torch's bundled HIP runtime has to come up
    # before the system one libmsiren links (mri_inr_amd/_lib.py:_init_torch_runtime_first)
    if any(it.get_closest_marker("gpu") for it in items):
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def nerr(a, ref):
    """max|a-ref| / max|ref| -- the parity norm of SURVEY.md §8(d); element-wise rtol is
    ill-posed here because outputs cross zero."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30))


def rms(a, ref):
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.sqrt(np.mean((a - ref) ** 2)))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
