"""pytest configuration: markers + shared fixtures."""
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # build the checker (CPU restatement; the unmodified reference into oracle/_ref when
    # /root/reference is present) BEFORE collection: test modules decide at import time whether
    # oracle/_ref exists
    from pssbam_testlib import build_oracle
    build_oracle()


@pytest.fixture(scope="session")
def oracle():
    from pssbam_testlib import Oracle
    return Oracle()
