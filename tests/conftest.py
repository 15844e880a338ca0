import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Small batches skip the FFT path's pair exclusion by default (it does not pay below a few thousand pairs); the GPU tests are
# small batches built to hit edge cases, and every one of them should go through the exclusion: they run with 'always' unless
# a test asks otherwise (tests/test_pair_exclusion.py compares the modes).
os.environ.setdefault("SUSHI_HIP_EXCLUSION", "always")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def golden_index():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "find_substream_index.json")) as f:
        return json.load(f)
