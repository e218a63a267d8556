import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test (still part of the default CPU suite unless deselected)")
    # the gradient tests differentiate the UNCAPPED reference step sequence on purpose (short protocols, compared with the checker);
    # the library's warning about that is asserted in tests/test_gpu_round3.py, not repeated in every report
    config.addinivalue_line("filterwarnings", "ignore:gradients w.r.t. the rate parameters through an UNCAPPED:RuntimeWarning")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def ion():
    """The product package (hyphenated directory name -> importlib)."""
    import importlib
    return importlib.import_module("neural-ode-ion-channels_amd")


@pytest.fixture(scope="session")
def gpu(ion):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible (there is no CPU fallback)")
    ion.capi.lib()  # fails loudly if libionode.so is missing
    return torch.device("cuda:0")
