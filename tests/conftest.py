import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must FAIL (not skip) when the HIP path is unavailable on a GPU box; on a CPU-only box they are
    # deselected by `-m "not gpu"`.
    pass
