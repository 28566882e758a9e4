#!/bin/bash
# Where the command's 14 GB of resident memory at 10^8 rows are (round 4): /proc/self/status and malloc_stats at the end of the run.
#   gpurun -- 'bash tools/history/cli_memory_probe.sh 100000000 > gpurun_out/r04/cli_memory.txt 2>&1'
n=${1:-100000000}
python - <<PY
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import bench, cli_e2e
from sctagger_amd import synth
bc = synth.make_barcodes(25000, seed=3)
os.makedirs("/tmp/sct_mem", exist_ok=True)
synth.write_barcodes_tsv("/tmp/sct_mem/sr.tsv", bc)
segs, offs = bench.make_batch($n, bc, 3)
print(cli_e2e.write_lr_file("/tmp/sct_mem/lr.tsv.gz", segs, offs, level=9))
PY
for i in 1 2; do
  SCT_CLI_TIMING=2 python scTagger.py match_trie -lr /tmp/sct_mem/lr.tsv.gz -sr /tmp/sct_mem/sr.tsv -mr 2 -t 16 -o /tmp/sct_mem/out.tsv.gz 2>&1 | grep -E "cli memory|cli timing: sub|streamcli timing|^Total|max mmap bytes" | cut -c1-260
  rm -f /tmp/sct_mem/out.tsv.gz
done
