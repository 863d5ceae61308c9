import os
import sys

import pytest

# PyTorch bundles its own HIP runtime; a process that first touches the GPU through the system's runtime (libezpz_amd.so
# links /opt/rocm's) and imports torch afterwards ends up with two runtimes, and torch then sees no device.  The tests
# that use torch (device tensors, torch.distributed) may run after tests that do not: load it first, whatever the selection.
try:
    import torch  # noqa: F401
except Exception:  # (CPU-only checks of the C ABI run without it)
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_case(name: str, filename: str = "problem.md") -> str:
    with open(os.path.join(GOLDEN, "test_cases", name, filename)) as f:
        return f.read()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.lib()
    return O
