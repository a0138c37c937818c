import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # A process that dies of SIGABRT (the HIP runtime gives up without a word on some queue errors) leaves only Python's
    # "Fatal Python error: Aborted" behind: with this aid loaded the C call stack of the thread that called abort() is
    # in the log too (tools/debug/abort_trace.c, built by __graft_entry__.build(); it hands on to whoever handled the
    # signal before).
    so = os.path.join(ROOT, "tools", "debug", "abort_trace.so")
    if os.path.exists(so):
        try:
            import ctypes
            ctypes.CDLL(so)
        except OSError:
            pass


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.load()
