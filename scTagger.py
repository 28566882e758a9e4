#!/usr/bin/env python3
"""Drop-in entry point: `scTagger.py match_trie -lr ... -sr ... [-mr 2] [-bl 16] [-t 16] [-o out.tsv.gz]`."""
import os
import threading


def _wake_the_device():
    """The first HIP call of a process costs 0.15 - 0.25 s (runtime and driver initialisation).  Make it now, on a thread of its
    own, while the interpreter is still importing numpy and reading the barcode file; the library is loaded through its C-ABI
    only (no CPU fallback exists: a missing library or device is reported by the sub-command itself)."""
    try:
        import ctypes
        lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sctagger_amd", "csrc", "libsctagger_hip.so")
        ctypes.CDLL(lib).sct_device_count()
    except Exception:
        pass


def _keep_freed_memory():
    """The command allocates and frees buffers of 4 - 32 MB on dozens of threads (inflated pieces, parsed blocks, rendered rows,
    gzip members).  glibc serves such sizes by mmap and returns them by munmap: every one a page-table change that interrupts all
    CPUs the process runs on (TLB shootdown) and fresh pages that fault in one by one — on a 256-core host the 100 M-row command
    spent more time there than decoding (reader 1.6 GB/s on 32 threads).  Keep freed memory in the allocator instead."""
    try:
        import ctypes
        libc = ctypes.CDLL(None)
        libc.mallopt(-3, 32 << 20)     # M_MMAP_THRESHOLD: everything up to 32 MiB (the most glibc allows) from the heaps
        libc.mallopt(-1, 0x7FFFFFFF)   # M_TRIM_THRESHOLD: do not give the top of a heap back
        libc.mallopt(-2, 64 << 20)     # M_TOP_PAD: grow the main heap in large steps
    except Exception:
        pass


def _relaunch_with_huge_pages():
    """Large inputs (match_trie, an -lr file of 200 MB or more): run the command in a child process whose allocator asks for
    transparent huge pages (GLIBC_TUNABLES=glibc.malloc.hugetlb=1 — glibc reads its tunables at process start only, and offers no
    call that does the same later).  The command touches 4 GB of heap per 10^8 rows: a million page faults, and a TLB entry per 4 KB
    for the decoders' and the compressor's windows; with huge pages 10^8 rows take 1.75 instead of 2.07 s on the GPU box (10^7 rows:
    no difference, so small inputs are not relaunched — the second interpreter costs 0.02 s).  Nothing has touched the GPU yet: this
    process only waits and passes the child's exit status on.
    (Huge-page-backed heap arrays and the GPU: the library never hands memory that is not page-locked to the HIP runtime — a worker's
    first blocks, parsed into ordinary arrays before its device context exists, go through the context's page-locked ring by memcpy —
    so no transparent huge page is ever pinned in place under a running kernel: DESIGN.md 5c.)"""
    import sys
    if "glibc.malloc.hugetlb" in os.environ.get("GLIBC_TUNABLES", "") or os.environ.get("SCT_NO_RELAUNCH") or "match_trie" not in sys.argv[1:2]:
        return
    try:
        at = [i for i, a in enumerate(sys.argv) if a in ("-lr", "--long-read-segments")][-1]
        if os.path.getsize(sys.argv[at + 1]) < int(os.environ.get("SCT_RELAUNCH_MIN_BYTES", 200 << 20)):
            return
    except (IndexError, OSError, ValueError):
        return
    import signal
    env = dict(os.environ)
    env["GLIBC_TUNABLES"] = ":".join(x for x in (env.get("GLIBC_TUNABLES"), "glibc.malloc.hugetlb=1") if x)
    if os.environ.get("SCT_CLI_TIMING"):
        sys.stderr.write("cli timing: relaunching with GLIBC_TUNABLES=%s\n" % env["GLIBC_TUNABLES"])
        sys.stderr.flush()
    pid = os.posix_spawn(sys.executable, [sys.executable] + sys.argv, env)
    for sig in (signal.SIGINT, signal.SIGTERM, signal.SIGHUP):
        signal.signal(sig, lambda s, frame: os.kill(pid, s))
    while True:
        try:
            _, status = os.waitpid(pid, 0)
            break
        except InterruptedError:
            continue
    code = os.waitstatus_to_exitcode(status)
    os._exit(code if code >= 0 else 128 - code)


_relaunch_with_huge_pages()
_keep_freed_memory()
threading.Thread(target=_wake_the_device, daemon=True).start()

from sctagger_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main()
    # The work is done and every output file is closed: leave at once.  Unwinding the interpreter (worker threads, the gzip pool)
    # and the HIP runtime (freeing a few GB of device buffers) costs 0.12 s of a command that takes 0.7 s for 10^7 rows.
    import sys
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)
