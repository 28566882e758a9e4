#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box. Usage: bash tools/profile.sh <tag> [extra bench args]
# 1) kernel trace + stats of the bench command (both loops: device end-to-end and resident; the CLI timing is left out — it
#    runs in a child process); 2..) PMC counters of the resident loop, each pass in its own run
#    (never combined with trace domains other than --kernel-trace, see the pool rules).
set -o pipefail
tag=${1:-r02}; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
echo "== kernel trace" | tee -a $R/$out/progress.log
# (without the sct_match-on-numpy leg: its ring route launches the same kernels over 1 M-row chunks, and the per-kernel averages of the
#  trace are compared with the bench line's 10 M-row launches)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -o trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cli --no-other-configs --no-pageable "$@" > $R/$out/trace_bench.json 2> $R/$out/trace.err || { tail -5 $R/$out/trace.err; exit 1; }
cat $R/$out/trace_bench.json
[ -n "$TRACE_ONLY" ] && exit 0
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVES_EQ_64 SQ_WAVES_LT_64" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  echo "== pmc $pass" | tee -a $R/$out/progress.log
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/$out/pmc_$name -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --mode resident --no-cpu-baseline --no-cli "$@" > $R/$out/pmc_$name.json 2> $R/$out/pmc_$name.err || { tail -5 $R/$out/pmc_$name.err; echo "(pass failed, continuing)"; }
done
ls -R $R/$out | head -50
