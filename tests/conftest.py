import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The engine leaves the side streams off for steps as small as the ones the tests run (resnet_engine.py: "Side streams");
# the suite forces them on so that the multi-stream paths stay the ones under test (the variants without them set
# IIF_NO_WGRAD_STREAM / IIF_NO_BWD_SIDE themselves, test_side_streams_follow_the_size_of_the_step removes the override).
os.environ.setdefault("IIF_SIDE_STREAMS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load
