"""Host-side sanitizers: the plain-C / C++ code that runs on the CPU — the TSV parser / renderer of the command
line (csrc/sct_hostio.c), the bit-parallel header and the seed-table builder shared with the kernels (sct_core.h,
sct_tables.h, through tests/host_core_check.cpp) and the oracle — built with AddressSanitizer + UndefinedBehaviorSanitizer
and driven by the existing suites.  CPU only: the device library cannot be sanitised on this pool."""
import os
import subprocess
import sys

from tests import util

ROOT = os.path.dirname(util.HERE)


def _asan_runtime():
    p = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_c_helpers_and_oracle_under_asan_ubsan():
    """tests/test_fastio.py, test_inflate.py (the DEFLATE decoder on valid, truncated and corrupted streams), test_oracle.py and
    the CPU part of test_fuzz.py once more, in a python that has libasan preloaded and loads the
    sanitizer builds of libsctagger_io / liboracle: any heap overrun, use after free or undefined operation aborts it."""
    rt = _asan_runtime()
    assert rt, "gcc's libasan.so not found"
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "sctagger_amd", "csrc"), "asan"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               SCT_IO_LIB=os.path.join(ROOT, "sctagger_amd", "csrc", "libsctagger_io_asan.so"),
               SCT_ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_asan.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_fastio.py"),
                        os.path.join(ROOT, "tests", "test_inflate.py"), os.path.join(ROOT, "tests", "test_oracle.py"),
                        os.path.join(ROOT, "tests", "test_fuzz.py"), "-m", "not gpu"],
                       env=env, capture_output=True, cwd=ROOT, timeout=1200)
    out = r.stdout.decode() + r.stderr.decode()
    assert r.returncode == 0 and "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
    assert " passed" in out


def test_core_header_and_table_builder_under_asan_ubsan():
    build = os.path.join(util.HERE, "_build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "host_core_check_asan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-o", exe, os.path.join(util.HERE, "host_core_check.cpp")])
    r = subprocess.run([exe, "8"], capture_output=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1"), timeout=1200)
    out = r.stdout.decode() + r.stderr.decode()
    assert r.returncode == 0 and out.startswith("OK") and "runtime error" not in out, out[-4000:]


def test_copy_pool_protocol_under_tsan_and_asan():
    """The copy pool of the stream seam's ring route (csrc/sct_copy_pool.h) runs the ring's protocol on the host — ways in, a device that
    takes a while, ways out whose last part hands the slot to its next user, calls cut short — under ThreadSanitizer (which found that
    a finished group was read after its waiter could be gone: round 5) and under ASan + UBSan (tests/host_pool_check.cpp)."""
    build = os.path.join(util.HERE, "_build")
    os.makedirs(build, exist_ok=True)
    for name, flags, rounds in (("tsan", ["-fsanitize=thread"], "60"), ("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], "120")):
        exe = os.path.join(build, "host_pool_check_" + name)
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fno-omit-frame-pointer"] + flags + ["-o", exe, os.path.join(util.HERE, "host_pool_check.cpp")])
        r = subprocess.run([exe, rounds], capture_output=True, timeout=1200, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1:abort_on_error=1"))
        out = r.stdout.decode() + r.stderr.decode()
        assert r.returncode == 0 and out.startswith("OK") and "Sanitizer" not in out and "runtime error" not in out, (name, out[-4000:])


def test_inflate_differential_fuzz_short_run():
    """tools/fuzz_inflate.c for 1 000 mutated streams under ASan + UBSan: the serial decoder (csrc/sct_inflate.c), the parallel driver
    (csrc/sct_pinflate.c: threads, chunking, hand-over) and zlib must agree on which streams are valid and on every byte of the valid
    ones; every structured input also goes through the fast compressor (csrc/sct_fdeflate.c) and back through zlib.  The recorded long
    run — 1.02 M streams — is profiles/r04_inflate_fuzz.log."""
    build = os.path.join(util.HERE, "_build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "fuzz_inflate_asan")
    subprocess.check_call(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
                           os.path.join(ROOT, "tools", "fuzz_inflate.c"), os.path.join(ROOT, "sctagger_amd", "csrc", "sct_inflate.c"),
                           os.path.join(ROOT, "sctagger_amd", "csrc", "sct_pinflate.c"), os.path.join(ROOT, "sctagger_amd", "csrc", "sct_fdeflate.c"), "-lz", "-lpthread"])
    r = subprocess.run([exe, "1000", "77"], capture_output=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1"), timeout=900)
    out = r.stdout.decode() + r.stderr.decode()
    assert r.returncode == 0 and "DONE 1000 streams" in out and "decoder ok / zlib not 0, zlib ok / decoder not 0" in out, out[-3000:]
    assert "verdict differs 0" in out and "piece CRCs wrong 0" in out and "COMPRESSOR 1000 inputs" in out and "identical 1000" in out, out[-3000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out
