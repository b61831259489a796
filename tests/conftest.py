import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# torch (used by the nccl / gloo tests and by the GPU fixtures) bundles its own HIP runtime: it has to be loaded BEFORE
# libdada2hip.so pulls in /opt/rocm's libamdhip64, otherwise the process holds two runtimes and torch.cuda reports no
# device.  The library itself never needs torch.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_c():
    """The plain-C restatement (oracle/libdada2oracle.so), built on demand."""
    from oracle import cport
    cport.lib()
    return cport


@pytest.fixture(scope="session")
def oracle_ref():
    """The reference compiled in place (oracle/_ref); only where it has been built."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return ref
