import os
import sys

import pytest

# the GPU tests read fqd_kernel_times to assert WHICH kernels ran (the route a call took): event pairs around every
# launch, off by default in the library
os.environ.setdefault("FQD_KERNEL_TIMERS", "1")

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
    return load_ref_vectors()


GOLDEN_FILES = ("ref_vectors.json.gz", "ref_vectors_long.json.gz")


def load_ref_vectors():
    """Both fixtures minted by tests/golden/make_golden.py, as one {"cases": {...}}."""
    import gzip
    import json
    cases = {}
    for name in GOLDEN_FILES:
        with gzip.open(os.path.join(ROOT, "tests", "golden", name)) as fh:
            cases.update(json.load(fh)["cases"])
    return {"cases": cases}
