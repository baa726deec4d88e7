import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _golden(name):
    import json
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kats():
    """hand-derived known-answer vectors (kat.json) + the EngineerData-shaped goldens both oracles agreed on when
    tools/gen_engineerdata_golden.py wrote them (engineerdata_small.json)"""
    return _golden("kat.json")["kats"] + _golden("engineerdata_small.json")["kats"]


@pytest.fixture(scope="session")
def map_ref_goldens():
    return _golden("engineerdata_small.json")["map_refs"]
