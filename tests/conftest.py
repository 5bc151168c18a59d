import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


def pkg():
    """The product package lives in a directory whose name is not an identifier."""
    import importlib

    return importlib.import_module("cat-generator_amd")


@pytest.fixture(scope="session")
def catgan():
    return pkg()
