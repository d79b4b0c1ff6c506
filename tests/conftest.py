import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")
    config.addinivalue_line("markers", "lab: experiments of the lab library on a GPU (run with `pytest -m lab`; not part of `-m gpu`)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
