import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu")


@pytest.fixture(scope="session")
def ctx():
    """One mh_ctx on cuda:0 for the GPU session (fails loudly if the HIP library
    is missing or there is no gfx950 device -- there is no CPU fallback)."""
    from moped_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def sift():
    from moped_amd import synth
    return synth.load_sift_fixture()
