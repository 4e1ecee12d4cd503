"""Runs the full-size property tests (tests/test_gpu_full_size.py: the SHT pair at width 384 on the 1-degree and 0.25-degree
grids, the 0.25-degree network at its real shape) outside pytest's collection of the whole suite: ``python tools/full_size_properties.py``."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.exit(pytest.main(["-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_full_size.py")]))
