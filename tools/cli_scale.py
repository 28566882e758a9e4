#!/usr/bin/env python3
"""The command at 10 M and 100 M rows: gz -> gz wall time by gzip level of the input (9 = what the reference's
producer writes, /root/reference/scTagger.py:377), inflate threads (1 = the serial decoder of round 3), workers, block size and
--devices.  The input files are written once per size and level; every variant runs the command twice (best reported).
  gpurun -- 'python tools/cli_scale.py 10000000 100000000 > gpurun_out/r04/cli_scale.jsonl'"""
import gzip, json, os, resource, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench, cli_e2e
from sctagger_amd import synth

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [10_000_000]
quick = "--quick" in sys.argv
work = "/tmp/sct_scale"
os.makedirs(work, exist_ok=True)
bc = synth.make_barcodes(25000, seed=3)
sr = os.path.join(work, "sr.tsv")
synth.write_barcodes_tsv(sr, bc)
ENVS = ("SCT_INFLATE_THREADS", "SCT_CLI_WORKERS", "SCT_STREAM_BLOCK_BYTES", "SCT_INFLATE_CHUNK_BYTES")
for n in sizes:
    t0 = time.time()
    segs, offs = bench.make_batch(n, bc, 3)
    t1 = time.time()
    files = {}
    for level in (9, 1):
        p = os.path.join(work, "lr_%d_l%d.tsv.gz" % (n, level))
        raw, gz = cli_e2e.write_lr_file(p, segs, offs, level=level)
        files[level] = (p, raw, gz)
    print(json.dumps({"rows": n, "batch_generated_s": round(t1 - t0, 1), "inputs_written_s": round(time.time() - t1, 1),
                      "raw_MB": round(files[9][1] / 1e6), "gz_MB": {l: round(files[l][2] / 1e6, 1) for l in files}}), flush=True)
    del segs
    B8, B16 = {"SCT_STREAM_BLOCK_BYTES": str(8 << 20)}, {"SCT_STREAM_BLOCK_BYTES": str(16 << 20)}
    variants = [(9, {}, []), (9, {"SCT_INFLATE_THREADS": "1"}, []), (9, {"SCT_INFLATE_THREADS": "12"}, []), (9, {"SCT_CLI_WORKERS": "8"}, []), (9, {}, ["--gzip-level", "4"]),
                (9, {}, ["--gzip-level", "1"]), (1, {}, []), (9, {}, ["--devices", "0", "0"]), (9, {}, ["-o", os.path.join(work, "out.tsv")])]
    if quick:
        variants = variants[:4]
    if "--thp" in sys.argv:      # malloc'ed memory in transparent huge pages (fewer faults on first touch, fewer pages to free at exit)
        T = {"GLIBC_TUNABLES": "glibc.malloc.hugetlb=1"}
        variants = [(9, {}, []), (9, T, []), (9, {"SCT_INFLATE_THREADS": "12"}, []), (9, dict(T, SCT_INFLATE_THREADS="12"), [])]
    out = os.path.join(work, "out.tsv.gz")
    rows_out = None
    for level, env, extra in variants:
        e = {k: v for k, v in os.environ.items() if k not in ENVS}
        e.update(env, SCT_CLI_TIMING="1")
        walls, cpu = [], []
        for _ in range(2):
            for stale in (out, os.path.join(work, "out.tsv")):     # a fresh output file, as in a first run (renaming over last run's
                try:                                               # 1.1 GB file costs 0.2 s of freeing it inside the rename)
                    os.remove(stale)
                except OSError:
                    pass
            r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
            t0 = time.time()
            rc = subprocess.call([sys.executable, os.path.join(ROOT, "scTagger.py"), "match_trie", "-lr", files[level][0], "-sr", sr, "-mr", "2", "-t", "16"] + (extra if "-o" in extra else ["-o", out] + extra),
                                 stderr=open(os.path.join(work, "cli.err"), "w"), env=e)
            walls.append(round(time.time() - t0, 3))
            r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
            cpu.append([round(r1.ru_utime - r0.ru_utime, 2), round(r1.ru_stime - r0.ru_stime, 2)])
            if rc != 0:
                break
        timing = [l.strip() for l in open(os.path.join(work, "cli.err")) if "timing" in l]
        if rows_out is None and rc == 0 and "-o" not in extra:        # once per size: every row came out (counted in pieces)
            rows_out = 0
            with gzip.open(out, "rb") as f:
                while True:
                    piece = f.read(64 << 20)
                    if not piece:
                        break
                    rows_out += piece.count(b"\n")
        print(json.dumps({"rows": n, "in_gzip_level": level, "in_gz_MB": round(files[level][2] / 1e6, 1), "env": env, "extra": extra, "rc": rc, "wall_s": walls, "cpu_user_sys_s": cpu,
                          "segments_per_s": n / min(walls), "out_MB": round(os.path.getsize(extra[extra.index("-o") + 1] if "-o" in extra else out) / 1e6, 1) if rc == 0 else None, "rows_out": rows_out,
                          "timing": timing[-5:]}), flush=True)
    for l in files:
        os.remove(files[l][0])
