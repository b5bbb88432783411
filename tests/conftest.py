import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # build the native pieces when a fresh checkout has not run __graft_entry__.build() yet (hipcc cross-compiles
    # without a GPU; on the GPU box the prebuilt .so files travel with the snapshot)
    import subprocess
    lib = os.path.join(ROOT, "phnn_mpc_amd", "csrc", "libphnn_mpc.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.check_call(["make", "-j2", "-C", os.path.dirname(lib), "-s"])


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib
