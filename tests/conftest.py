import os
import sys

import pytest

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
