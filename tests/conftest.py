import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The library is built in-tree ahead of the tests (`__graft_entry__.build()`); it is git-ignored, so a fresh
    # checkout has none.  Build it once then, rather than fail every test that imports `lcm` (a no-op otherwise).
    lib = os.environ.get("LCMHIP_LIB") or os.path.join(ROOT, "chimera-llm_amd", "lcm", "liblcm_hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def normalizer():
    import oracle
    return oracle.load_dummy_normalizer()
