import sys
from pathlib import Path

import pytest

import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # as the package does on import (before HIP initialises)
ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
