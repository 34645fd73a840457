import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stvo-pl_amd", "python"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def hip():
    """The C-ABI product library on a real GPU.  Fails loudly (no CPU fallback) if unusable."""
    from stvo_amd import capi
    lib = capi.load()
    ctx = capi.Context(device_id=0, max_rows=4096, max_batch=64)
    yield ctx
    ctx.close()


@pytest.fixture
def switches(monkeypatch):
    """Developer switches of the library (stvo-pl_amd/csrc/debug_switches.h): the STVO_* variables are parsed once, so a test
    that drives a variant sets them through this fixture, which makes the library read the environment again — and once more
    after the test has restored it."""
    from stvo_amd import capi

    def set_env(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        capi.load().stvo_debug_reparse_env()

    yield set_env
    monkeypatch.undo()
    capi.load().stvo_debug_reparse_env()
