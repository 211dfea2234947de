import os
import sys

sys.dont_write_bytecode = True      # tests import the read-only reference checkout (tests/helpers/ref_host_boundary.py): no __pycache__ there

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def ctx():
    """el_ctx on cuda:0 -- fails (does not skip) when the HIP library or the GPU is missing."""
    from elliot_amd import ops
    return ops.get_context(0)


@pytest.fixture(scope="session")
def dev(ctx):
    return ctx.device


@pytest.fixture
def lib_option(ctx):
    """lib_option(name, value): a switch of the library (el_ctx_set_option) for the duration of one test."""
    saved = []

    def set_(name, value):
        saved.append((name, ctx.set_option(name, value)))
    yield set_
    for name, old in reversed(saved):
        ctx.set_option(name, old)
