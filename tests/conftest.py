import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_libraries():
    """The shared libraries are built in-tree (and git-ignored); build them once if a fresh checkout lacks them."""
    from sctagger_amd import _native, fastio
    from oracle import trie_oracle
    if not (os.path.exists(_native.LIB_PATH) and os.path.exists(fastio.IO_LIB_PATH) and os.path.exists(trie_oracle.LIB_PATH)):
        import __graft_entry__
        __graft_entry__.build()
