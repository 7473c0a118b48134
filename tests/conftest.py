import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def ctx():
    """One zkp_ctx on cuda:0.  Fails loudly (no skip, no fallback) if the HIP library or the device is missing."""
    from ckb_zkp_amd.api import Context
    c = Context(int(os.environ.get("ZKP_TEST_DEVICE", "0")))
    yield c
    c.close()
