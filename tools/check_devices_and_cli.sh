#!/bin/bash
# GPU-box check: the threaded --devices path (several contexts on one GPU) and the CLI end-to-end timing.
set -e
g=tests/golden
python scTagger.py match_trie -lr $g/g2.lr.tsv -sr $g/g2.sr.tsv -mr 2 -t 5 -o /tmp/o1.tsv.gz --devices 0 0 0 2>/dev/null
python - <<'PY'
import gzip
a=gzip.open('/tmp/o1.tsv.gz','rb').read(); b=open('tests/golden/g2.e2.t5.expected.tsv','rb').read()
print("devices 0 0 0 ->", "identical" if a==b else "DIFFERENT", len(a))
PY
python tools/cli_e2e.py 10000000 /tmp/sct_e2e 2>&1 | tail -1
