import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU-marked tests need a real device: on a CPU box they are skipped, not failed (plain `pytest` works here)."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """tests/golden/<name>.npz -> {case: {key: ndarray}} (minted by oracle/mint_goldens.py from the
    reference's own source lines)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cases = {}
    for k in z.files:
        case, key = k.split("/", 1)
        cases.setdefault(case, {})[key] = z[k]
    return cases


@pytest.fixture(scope="session")
def golden():
    return load_golden
