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


_SYNTH = {}


def load_case(name):
    if name == "case9241synth":                      # seeded generator (juliagrid.jl_amd/synthetic.py), cached per session
        if name not in _SYNTH:
            from juliagrid.jl_amd.synthetic import case9241synth
            _SYNTH[name] = case9241synth()
        return {k: np.array(v) for k, v in _SYNTH[name].items()}
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
    # torch ships its own copy of the HIP runtime under the same soname: when a process uses both torch and libjgrid_hip.so,
    # torch has to be imported first so that both bind to ONE runtime (bench.py does the same); loading ours first and torch
    # later leaves torch without devices ("No HIP GPUs are available").
    if not os.environ.get("JG_PLAN_LIB"):            # (a sanitizer run of the plan code preloads libasan: torch does not survive that)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    import juliagrid.jl_amd as jg
    return jg
