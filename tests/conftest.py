import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_bind
    return oracle_bind.load()


@pytest.fixture(scope="session")
def gpu_ctx():
    import path_tracer_amd as pt
    ctx = pt.Context(0)
    yield ctx
    ctx.close()
