import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # The CPU oracle and the host library build in seconds with g++; the HIP library is
    # cross-compiled by __graft_entry__.build() and is only rebuilt here when it is missing.
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    if not (os.path.exists(os.path.join(ROOT, "raytracer_amd", "librt_hip.so"))
            and os.path.exists(os.path.join(ROOT, "raytracer_amd", "librt_host.so"))):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "raytracer_amd", "csrc")])


@pytest.fixture(scope="session")
def root():
    return ROOT
