#!/bin/bash
# Round 3's committed profiles: the bench command under rocprofv3 (kernel trace + stats), the PMC passes of the resident loop at
# cfg3, and the same PMC passes at cfg5 (10 M rows x 3.6 M barcodes, full size).  Run on the GPU box; then
#   python tools/summarize_prof.py gpurun_out/prof_r03 r03 && python tools/summarize_prof.py gpurun_out/prof_r03cfg5 r03cfg5 k_filter_counters_cfg5.json
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/profile.sh r03 || exit 1
bash $R/tools/profile.sh r03cfg5 --workload cfg5 --no-cpu-baseline || exit 1
