import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle (torch-CPU) is the checker of most GPU tests.  On a 128-thread host shared with other jobs an OpenMP team of 128
    # busy-waiting threads can collapse (one validation run of round 2 sat 20+ minutes in a 33-second oracle call): 32 threads are within
    # 2x of the best time for every oracle call of the suites and far less exposed.
    try:
        import torch
        torch.set_num_threads(min(32, torch.get_num_threads()))
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
        return cache[name]

    return load
