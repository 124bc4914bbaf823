import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def pha():
    import phantom_fhe_amd
    return phantom_fhe_amd


@pytest.fixture(scope="session")
def gpu(pha):
    """Loads the HIP library (fails loudly when it is missing) and checks a device is there."""
    import torch
    assert os.path.exists(pha.LIB_PATH), "libphantom_amd.so missing on the GPU box: build it in-tree first"
    pha.lib()
    # key material is compared with the oracle bit for bit: both sides sample under the oracle's fixed key (the
    # library's own default is 32 bytes of OS entropy; tests/test_keygen.py checks that default too)
    import oracle
    pha.set_sampler_key(oracle.SAMPLER_KEY)
    assert torch.cuda.is_available(), "GPU test selected but no HIP device is visible"
    torch.cuda.set_device(0)
    return torch
