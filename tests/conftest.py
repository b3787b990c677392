import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    path = os.path.join(ROOT, "tests", "golden", "numpy_stages.npz")
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


GOLDEN_TAGS = ["uniform_48x64", "hazy_37x53", "greenish_40x56", "bluish_40x56"]
