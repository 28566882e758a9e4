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
