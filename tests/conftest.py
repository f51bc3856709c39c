import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_runtest_logfinish(nodeid, location):
    """flush the progress output after every test: with stdout redirected to a file pytest's dots sit in a block buffer,
    and a long GPU run then looks silent (the GPU box's watchdog kills commands that write nothing for minutes)"""
    tr = _CONFIG.pluginmanager.get_plugin("terminalreporter") if _CONFIG is not None else None
    if tr is not None:
        try:
            tr._tw.flush()
        except Exception:
            pass


_CONFIG = None


@pytest.hookimpl(trylast=True)
def pytest_sessionstart(session):
    global _CONFIG
    _CONFIG = session.config
