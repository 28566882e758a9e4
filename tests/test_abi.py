"""The C-ABI library builds for gfx950, loads, and exports every symbol include/sctagger_hip.h declares.
No compute is attempted here (no GPU); on a box without a device the library must fail loudly."""
import ctypes
import os
import re

import pytest

from sctagger_amd import _native
from tests import util

ROOT = os.path.dirname(util.HERE)


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    if not os.path.exists(_native.LIB_PATH):
        __graft_entry__.build()
    return _native.load()


def declared_symbols():
    with open(os.path.join(ROOT, "include", "sctagger_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sct_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    syms = declared_symbols()
    assert set(syms) == set(_native.EXPORTS)
    for s in syms:
        assert hasattr(lib, s), "libsctagger_hip.so does not export %s" % s


def test_prefix_len(lib):
    # reference scTagger.py:689-697: smallest k with 4^k >= threads
    assert [_native.prefix_len(t) for t in (1, 2, 4, 5, 16, 17, 64, 65)] == [0, 1, 1, 2, 2, 3, 3, 4]


def test_fails_loudly_without_a_device(lib):
    if _native.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_native.SctError) as ei:
        _native.Matcher(0)
    assert ei.value.code == -3
    assert "no CPU fallback" in str(ei.value)


def test_product_does_not_reference_the_oracle():
    """The shipped package must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "sctagger_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                with open(os.path.join(dirpath, fn)) as f:
                    assert "oracle" not in f.read().lower(), os.path.join(dirpath, fn)
    out = os.popen("ldd %s" % _native.LIB_PATH).read()
    assert "oracle" not in out
