import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The oracle (and the reference build) are OpenMP code, and libgomp sizes its teams by the CPUs it SEES: a GPU box
# shows 256 of them behind a 16-core quota, and 256 spinning threads on 16 cores' worth of time turn a 0.1 s
# oracle run into minutes (seen: a test stuck at 1 500 % CPU with 321 threads).  Read once, when libgomp loads --
# so here, before any test imports a library.  (The product's own host loops ask for at most 8 threads.)
os.environ.setdefault("OMP_NUM_THREADS", str(min(8, os.cpu_count() or 8)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "refprobe: needs oracle/_ref (build container only)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import sift3d_oracle as so
    so.lib()
    return so
