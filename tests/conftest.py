import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout(seconds): per-test limit (pytest-timeout; a no-op without the plugin)")


def pytest_collection_modifyitems(config, items):
    """No test of this suite needs more than a few seconds of GPU time: a hung child process or device must fail ONE
    test after ten minutes, not eat the whole run (pytest-timeout, when installed)."""
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def s3r():
    import s3r as pkg
    return pkg


@pytest.fixture(scope="session")
def oracle():
    from oracle import s2v_oracle
    return s2v_oracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
