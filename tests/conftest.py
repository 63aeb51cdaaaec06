import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rtx():
    """The product package (directory name has a hyphen)."""
    return importlib.import_module("rust-raytracer_amd")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    mod = importlib.import_module("oracle.oracle")
    mod.build()
    mod.lib()
    return mod
