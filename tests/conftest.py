import os
import subprocess
import sys

import pytest

# The product library compiles kernels for a model's own shape at nepmi_model_load when it carries none (capi_jit.h: ~50 s of
# hipcc per shape).  The suite's run-time-shape cases (the shipped Si / C-2024 models, water ...) are there to test the
# run-time-shape kernels, and a GPU box starts with an empty cache: off by default; tests/test_jit_shapes.py asks for it.
os.environ.setdefault("NEPMI_JIT", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_checkers():
    """Build the CPU checkers (oracle; reference NEP_CPU where /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    if os.path.isdir("/root/reference"):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref"], check=True)
    yield
