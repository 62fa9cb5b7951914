import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tiny_scene():
    from agile_grasp_amd import synthetic

    return synthetic.config("tiny")


@pytest.fixture(scope="session")
def small_scene():
    from agile_grasp_amd import synthetic

    return synthetic.config("small")


@pytest.fixture(scope="session")
def svm_model():
    import numpy as np

    z = np.load(os.path.join(ROOT, "tests", "golden", "svm_weights.npz"))
    return z["w"], float(z["rho"])
