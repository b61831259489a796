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


def pytest_collection_modifyitems(config, items):
    """The at-size tests go LAST, and the reference runs they compare with are started in background processes right away
    (tests/at_size.py): the rest of the GPU suite runs while the host's cores work through them."""
    sized = [it for it in items if "at_size" in it.nodeid or "headline_1M" in it.nodeid]
    if not sized or "not gpu" in (config.getoption("-m") or "") or config.getoption("--collect-only"):
        return
    rest = [it for it in items if it not in sized]
    items[:] = rest + sized
    try:
        import torch
        from oracle import ref
        if torch.cuda.is_available() and ref.available():
            import at_size
            want = [c for c, key in (("cfg3", "headline_1M"), ("cfg4", "config4"), ("cfg5", "at_size_config5")) if any(key in it.nodeid for it in sized)]
            at_size.start(tuple(want))
    except Exception:   # pragma: no cover  (no GPU / no reference here: the tests skip themselves)
        pass


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
