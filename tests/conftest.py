import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    # torch bundles its own HIP runtime: when a test uses torch.cuda (RCCL) in the same process as the engine's library,
    # torch has to touch the GPU first, or it no longer finds one.  No-op on a machine without a GPU.
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # noqa: BLE001 - torch is optional for everything but the RCCL tests
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    d = np.load(os.path.join(ROOT, "tests", "golden", "mpc_golden.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def ospec(golden):
    from oracle.mpc_nlp import MpcSpec

    return MpcSpec(A_obs=golden["A_obs"], b_obs=golden["b_obs"], n_nbr=3)
