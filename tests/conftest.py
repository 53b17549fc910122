import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; see oracle/fqd_oracle.h)."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def known_answers():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def ref_vectors():
    import gzip
    import json
    with gzip.open(os.path.join(ROOT, "tests", "golden", "ref_vectors.json.gz")) as fh:
        return json.load(fh)
