import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("CANVAS_TEST_HOOKS", "1")      # the library reads its CANVAS_* test / diagnostic switches only with this set (common.hpp: cvx_hook)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")    # CBS keeps more than 4 kernels in flight (canvas_amd/__init__.py: the host application sets this, not the package)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
