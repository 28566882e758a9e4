"""The bit-parallel arithmetic of sctagger_amd/csrc/sct_core.h (Myers step, transposed Wu-Manber, the
six seed families) compiled for the HOST and checked against a plain DP — no GPU needed."""
import os
import subprocess

from tests import util


def test_core_header_on_host():
    build = os.path.join(util.HERE, "_build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "host_core_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(util.HERE, "host_core_check.cpp")])
    out = subprocess.check_output([exe]).decode()
    assert out.startswith("OK"), out


def test_copy_pool_protocol_on_host():
    """csrc/sct_copy_pool.h alone: the ring route's protocol (tests/host_pool_check.cpp), pools of 0 .. 6 threads, 1 500 calls."""
    build = os.path.join(util.HERE, "_build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "host_pool_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(util.HERE, "host_pool_check.cpp")])
    out = subprocess.check_output([exe, "300"]).decode()
    assert out.startswith("OK"), out
