import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sdfa-2019_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class G:
        def __getitem__(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"))
    return G()


@pytest.fixture(scope="session")
def synth_sd():
    from sdfa_amd import synth
    return {h: synth.make_state_dict(h, 1234) for h in ("dgrad", "offsets")}
