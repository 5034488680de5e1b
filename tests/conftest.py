import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "cuda-path-tracer_amd", "python"))
sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_api

    oracle_api.lib()
    return oracle_api


@pytest.fixture(scope="session")
def pt():
    import ptamd

    ptamd.lib()
    return ptamd


@pytest.fixture(scope="session")
def scenes():
    from ptamd import scenes as s

    return s


@pytest.fixture(scope="session")
def teapot(scenes):
    pos, tri, uv, vn = scenes.load_obj(os.path.join(REPO, "examples", "teapot", "utah.obj"))
    return pos, tri
