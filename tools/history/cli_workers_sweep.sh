cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, time, subprocess, json
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import cli_e2e, bench
from sctagger_amd import synth
bc = synth.make_barcodes(25000, seed=3)
segs, offs = bench.make_batch(10_000_000, bc, 3)
work = "/tmp/sct_t"; os.makedirs(work, exist_ok=True)
lrgz = os.path.join(work, "lr.tsv.gz"); cli_e2e.write_lr_file(lrgz, segs, offs)
import gzip
data = gzip.open(lrgz, "rb").read(); open(os.path.join(work, "lr.tsv"), "wb").write(data); del data
sr = os.path.join(work, "sr.tsv"); synth.write_barcodes_tsv(sr, bc)
for inp, w, blk in (("lr.tsv", "8", 4), ("lr.tsv", "4", 4), ("lr.tsv", "4", 2), ("lr.tsv.gz", "4", 4), ("lr.tsv.gz", "8", 4), ("lr.tsv.gz", "4", 2)):
    if True:
        env = dict(os.environ, SCT_CLI_WORKERS=w, SCT_CLI_TIMING="1", SCT_STREAM_BLOCK_BYTES=str(blk << 20))
        best = 9; err = b""
        for _ in range(2):
            t0 = time.time()
            r = subprocess.run([sys.executable, "scTagger.py", "match_trie", "-lr", os.path.join(work, inp), "-sr", sr, "-mr", "2", "-t", "16", "-o", os.path.join(work, "out.tsv.gz")], capture_output=True, env=env)
            dt = time.time() - t0
            if dt < best: best, err = dt, r.stderr
        print(inp, "workers", w, "block MB", blk, "wall %.3f" % best, flush=True)
        for l in err.decode().splitlines():
            if "timing" in l:
                print("    " + l, flush=True)
t0=time.time(); subprocess.run([sys.executable, "-c", "import numpy, ctypes; import sctagger_amd.cli"]); print("python+imports", round(time.time()-t0,3))
PY
