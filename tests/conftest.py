import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def golden_tables():
    """Schedule tables exactly as the reference computed them on the host that
    produced tests/golden (torch.sqrt is host-dependent by 1 ulp, see
    tinydiffusionmodels_amd/schedule.py:set_tables)."""
    import numpy as np
    import torch
    z = np.load(os.path.join(GOLDEN, "schedule.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}
