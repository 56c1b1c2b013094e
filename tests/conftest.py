import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    here = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(here, "adaptive_static_kat.json")) as f:
        kat = json.load(f)
    with open(os.path.join(here, "blocks.json")) as f:
        blocks = json.load(f)["blocks"]
    with open(os.path.join(here, "long_streams.json")) as f:
        long_streams = json.load(f)
    with open(os.path.join(here, "rans.json")) as f:
        rans = json.load(f)
    return {"kat": kat, "blocks": blocks, "long": long_streams, "rans": rans}


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build_oracle()
    return oracle_lib.oracle()


@pytest.fixture(scope="session")
def reference():
    """The real reference build; None outside the build container unless oracle/_ref travelled."""
    import oracle_lib
    return oracle_lib.reference()
