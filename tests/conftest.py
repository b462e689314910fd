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


def load_csv(name):
    # the reference's CSVs hold floats such as "5296674829675476e-20": plain float() parses them
    return np.loadtxt(os.path.join(GOLDEN, name), delimiter=",", ndmin=2)


@pytest.fixture(scope="session")
def golden():
    return load_csv


# the reference's test models (test/common.jl:15-32)
MODELS = {
    "a": np.array([[0.0, 0.1, 0.2], [0.1, 0.0, 0.3], [0.2, 0.3, 0.0]]),
    "b": np.array([[0.3, 0.1, 0.2], [0.1, 0.2, 0.3], [0.2, 0.3, 0.1]]),
    "c": np.array([[0.0, 0.1, 0.2, 0.3], [0.1, 0.0, 0.2, 0.3], [0.2, 0.2, 0.0, 0.3], [0.3, 0.3, 0.3, 0.0]]),
}
# default regularisers (GraphicalModelLearning.jl:35,49,56)
DEFAULT_C = {"RISE": 0.4, "logRISE": 0.8, "RPLE": 0.2}
