"""pytest wiring.  `-m "not gpu"`: oracle pins, host logic, ABI surface (runs without a GPU).
`-m gpu`: parity of the HIP path against the oracle, through the C ABI, on a real MI355X."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
SEED = 0xA11CE


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure)"""
    o = entry.load_oracle()
    o.lib()
    return o


@pytest.fixture(scope="session")
def pkg():
    """the product package; building needs hipcc but no GPU"""
    return entry.build()


@pytest.fixture(scope="session")
def ctx(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but no HIP device is visible (there is no CPU fallback)")
    c = pkg.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def amv1(orc):
    """the reference's own fixture C-AMVDecoder/bin/AMV1.amv (data file, committed under tests/golden)"""
    data = open(os.path.join(GOLDEN, "AMV1.amv"), "rb").read()
    info, vids, auds = orc.parse_amv(data)
    return {"data": data, "info": info, "video": vids, "audio": auds, "path": os.path.join(GOLDEN, "AMV1.amv")}
