"""`scTagger.py match_trie` (and `extract_sr_bc_from_lr`) command line (reference scTagger.py:67-97, :135-143, :847-861).

Flags, defaults and asserts are the reference's for this sub-command; `-m/--mem` is accepted and unused, exactly
as there; `-p/--plotfile` draws the distance histogram the reference's README advertises.  Optional extras: `--devices` (GPU ids, one shard each) and
`--gzip-level`.  The other scTagger sub-commands are outside this build's scope.
"""
import argparse
import sys

from . import match_trie as _mt

OTHER_SUBCOMMANDS = ("extract_sr_bc",)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="scTagger match_trie on AMD MI355X")
    subparsers = parser.add_subparsers(dest="subcommand")
    subparsers.required = True
    p = subparsers.add_parser("match_trie")
    p.add_argument("-lr", "--long-read-segments", type=str, required=True, help="Long-read segments TSV file")
    p.add_argument("-sr", "--short-read-barcodes", type=str, required=True, help="Short-read barcode list TSV file")
    p.add_argument("-mr", "--max-error", default=2, type=int,
                   help="Maximum number of errors allowed for barcode matching. Default: 2")
    p.add_argument("-m", "--mem", default=16.0, type=float, help="Accepted for compatibility; unused. Default: 16.0")
    p.add_argument("-bl", "--barcode-length", default=16, type=int, help="Length of barcodes. Default: 16")
    p.add_argument("-t", "--threads", default=16, type=int,
                   help="Reference thread count; here it only selects the reference's window-prefix length "
                        "(affects segments with non-ACGT bytes). Default: 16")
    p.add_argument("-p", "--plotfile", default=None, type=str, help="Path of plot file (histogram of the minimum edit distance). Default: no plotting")
    p.add_argument("-o", "--outfile", type=str, default=None,
                   help="Path to output file; gzipped if it ends with 'gz'. Default: stdout")
    p.add_argument("--devices", type=int, nargs="+", default=None, help="GPU ids to shard the segments over. Default: 0")
    p.add_argument("--no-stream", action="store_true", help="read the whole -lr file before matching, as the reference does, instead of streaming it in blocks")
    p.add_argument("--gzip-level", type=int, default=-1,
                   help="how the output is deflated: 1..9 = zlib at that level (the reference writes level 9); -1 = the command's own fast "
                        "compressor (three times zlib level 4's speed, a slightly smaller file). The bytes after decompression are identical. Default: -1")
    experimental = ("EXPERIMENTAL: the reference does this step with a third-party library that is not available to this build "
                    "(%s); the results follow that library's documented behaviour and could not be compared with it")
    l = subparsers.add_parser("extract_lr_bc", description=experimental % "edlib", help="experimental")   # reference scTagger.py:25-49
    l.add_argument("-r", "--reads", nargs="+", type=str, required=True, help="Space separated paths to reads in FASTQ")
    l.add_argument("-g", "--ranges", nargs="+", type=str, default=list(),
                   help="Ranges where the SR adapter should be found, e.g. f20:40 r1:30 (1-indexed, inclusive). Default: detect from data")
    l.add_argument("-z", "--gzipped", dest="gzipped", action="store_true", help="Input is gzipped (default: by the .gz suffix)")
    l.add_argument("-t", "--threads", default=1, type=int, help="Accepted for compatibility; must be positive. Default: 1")
    l.add_argument("-sa", "--short-read-adapter", type=str, default="CTACACGACGCTCTTCCGATCT",
                   help="Short-read adapter. Default: CTACACGACGCTCTTCCGATCT")
    l.add_argument("-o", "--outfile", type=str, default=None, help="Path to output file (gzipped). Default: stdout")
    l.add_argument("-p", "--plotfile", type=str, default=None, help="Path to plot file. Default: no plotting")
    l.add_argument("--num-bp-after", type=int, default=20, help="Bases after the end of the adapter alignment to output. Default: 20")
    q = subparsers.add_parser("extract_sr_bc_from_lr", description=experimental % "pyahocorasick", help="experimental")   # reference scTagger.py:67-79
    q.add_argument("-i", "--input", type=str, required=True, help="Input TSV file generated from extract_lr_bc step")
    q.add_argument("-o", "--outfile", type=str, default=None, help="Path to output file (gzipped). Default: STDOUT")
    q.add_argument("-wl", "--barcode-whitelist", type=str, required=True, help="Path to TXT barcode whitelist")
    q.add_argument("--thresh", type=float, default=0.005)
    q.add_argument("--step-size", type=int, default=1000)
    q.add_argument("--max-barcode-cnt", type=int, default=25_000)
    for name in OTHER_SUBCOMMANDS:
        subparsers.add_parser(name, add_help=False)
    args, rest = parser.parse_known_args(argv)
    if args.subcommand in OTHER_SUBCOMMANDS:
        parser.error("sub-command %r is not part of this build" % args.subcommand)
    if rest:
        parser.error("unrecognized arguments: %s" % " ".join(rest))
    if args.subcommand == "extract_lr_bc":           # reference :100-101, :128 (the -g syntax is checked when it is parsed)
        assert 0 < args.num_bp_after
        assert args.threads > 0
    if args.subcommand == "extract_sr_bc_from_lr":   # reference :135-138
        assert 0 <= args.thresh <= 1
        assert 0 < args.step_size
        assert 0 < args.max_barcode_cnt
    if args.subcommand == "match_trie":              # reference :140-143
        assert args.mem > 0
        assert args.barcode_length > 0
        assert args.barcode_length > args.max_error >= 0
    return args


def since_process_start():
    """Seconds since this process was started (Linux), for SCT_CLI_TIMING."""
    try:
        import os
        with open("/proc/self/stat") as f:
            start_ticks = int(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as f:
            up = float(f.read().split()[0])
        return up - start_ticks / os.sysconf("SC_CLK_TCK")
    except Exception:
        return -1.0


def main(argv=None):
    import os
    timing = bool(os.environ.get("SCT_CLI_TIMING"))
    if timing:
        sys.stderr.write("cli timing: main() entered %.3f s after process start\n" % since_process_start())
    args = parse_args(argv)
    print(args, file=sys.stderr)
    try:
        _dispatch(args)
    finally:
        if timing:
            sys.stderr.write("cli timing: sub-command returned %.3f s after process start\n" % since_process_start())
            if os.environ.get("SCT_CLI_TIMING") == "2":      # where the memory is: the kernel's view and the allocator's
                try:
                    with open("/proc/self/status") as f:
                        sys.stderr.write("cli memory: " + " ".join(l.strip().replace("\t", "") for l in f if l.startswith(("VmHWM", "VmRSS", "RssAnon", "RssFile", "RssShmem"))) + "\n")
                    import ctypes
                    sys.stderr.flush()
                    ctypes.CDLL(None).malloc_stats()
                except Exception:
                    pass


def _dispatch(args):
    if args.subcommand == "extract_lr_bc":
        from . import extract_lr_bc as _l
        _l.extract_lr_bc(args)
    elif args.subcommand == "extract_sr_bc_from_lr":
        from . import extract_sr_bc_from_lr as _x
        _x.extract_sr_bc_from_lr(args)
    else:
        _mt.match_trie(args)


if __name__ == "__main__":
    main()
