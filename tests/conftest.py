import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = os.path.join(GOLDEN, "cases")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_case(name):
    with np.load(os.path.join(CASES, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def load_golden(name):
    with np.load(os.path.join(GOLDEN, f"results_{name}.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def jg():
    import juliagrid.jl_amd as jg
    return jg
