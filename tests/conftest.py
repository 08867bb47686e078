import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. `pytest tests` in the build container."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library is a build product (git-ignored): a fresh checkout has none.  Tests need it — the CPU suite to check the
    exported symbols and the host-side argument checks, the GPU suite for everything — so it is built once here when missing
    (hipcc cross-compiles without a GPU; ~2 minutes).  The product itself never builds implicitly: it raises VfnError."""
    lib_path = os.path.join(REPO, "vf_nerf_amd", "csrc", "libvfn.so")
    if not os.path.exists(lib_path) and not os.environ.get("VFN_LIB"):
        import __graft_entry__
        __graft_entry__.build()
    yield
