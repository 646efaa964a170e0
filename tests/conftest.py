import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The built libraries are git-ignored: a fresh checkout has none.  Build them once (hipcc cross-compiles without a GPU);
    # this is the same make the driver's build() runs -- the product itself still refuses to run without its extension.
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "quadruped-gym_amd", "csrc", "libquadgym.so")
    if not os.path.exists(lib) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.run(["make", "-C", os.path.dirname(lib), "libquadgym.so"], check=False, capture_output=True)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture()
def model(oracle):
    return oracle.default_model()


@pytest.fixture()
def task(oracle):
    return oracle.default_task()
