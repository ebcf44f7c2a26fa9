import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def c_oracle():
    from oracle import c_oracle as c
    c.lib()
    return c


@pytest.fixture(scope="session")
def txctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ofdm_amd.batch import TxContext
    ctx = TxContext(0)
    yield ctx
    ctx.close()
