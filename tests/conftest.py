import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): oracle/flashe_oracle.{c,py}."""
    from oracle import flashe_oracle
    flashe_oracle.build()
    # many-core cloud hosts gain nothing past a few threads for these sizes (and lose to barriers)
    flashe_oracle.set_num_threads(min(os.cpu_count() or 1, 16))
    return flashe_oracle


def unhex(lst):
    return [int(x, 16) for x in lst]
