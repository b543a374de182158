import os
import sys

import numpy as np
import pytest
import scipy.sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
RUN_CASES = ["env75", "env300", "env192", "er120", "dense60"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def csr_from(g, prefix):
    shape = tuple(int(x) for x in g[prefix + "_shape"])
    return scipy.sparse.csr_matrix((g[prefix + "_data"], g[prefix + "_indices"], g[prefix + "_indptr"]), shape=shape)


def state_from(g, prefix=""):
    return csr_from(g, prefix + "S"), csr_from(g, prefix + "Q"), np.array(g[prefix + "h_max"])


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.linalg.norm((a - b).ravel())
    n = np.linalg.norm(b.ravel())
    return d / n if n > 0 else d


@pytest.fixture(params=RUN_CASES)
def run_case(request):
    return request.param, load_golden("run_" + request.param)
