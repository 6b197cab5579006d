import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip("golden fixture %s missing" % name)
    return np.load(path, allow_pickle=False)


def golden_input(name):
    return os.path.join(GOLDEN, "inputs", name + ".inp")


SMALL_CASES = ["bsp0", "c1_exp", "c1_lin", "rogers", "simfues", "bc1", "ka_ra", "lin256", "yuk256",
               "tiny8", "n65_k4", "n128", "bc10"]      # round 2: shapes at the edges of the kernels' tilings


def ulp_diff(a, b):
    """max |a-b| in units of the last place of b (elementwise spacing)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    sp = np.spacing(np.maximum(np.abs(a), np.abs(b)))
    sp[sp == 0] = np.finfo(float).tiny
    return float(np.max(np.abs(a - b) / sp)) if a.size else 0.0
