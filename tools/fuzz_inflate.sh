#!/bin/bash
# Differential fuzz run of csrc/sct_inflate.c (serial) and csrc/sct_pinflate.c (parallel driver) against zlib and against each other under ASan + UBSan (CPU only).  Usage: bash tools/fuzz_inflate.sh [streams per process] [processes]
# Writes profiles/r04_inflate_fuzz.log (one DONE line per process + the command); any sanitizer report or mismatch ends a process with a non-zero status.
set -o pipefail
per=${1:-300000}; procs=${2:-4}
R=$(cd "$(dirname "$0")/.." && pwd)
gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -o /tmp/fuzz_inflate $R/tools/fuzz_inflate.c $R/sctagger_amd/csrc/sct_inflate.c $R/sctagger_amd/csrc/sct_pinflate.c $R/sctagger_amd/csrc/sct_fdeflate.c -lz -lpthread || exit 1
log=$R/profiles/${FUZZ_LOG:-r05_inflate_fuzz.log}
{ echo "# tools/fuzz_inflate.sh $per $procs   ($(date -u +%Y-%m-%dT%H:%MZ), $(gcc --version | head -1), zlib $(python3 -c 'import zlib; print(zlib.ZLIB_VERSION)'))"
  echo "# gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined; each process: /tmp/fuzz_inflate $per <seed>"; } > $log
pids=()
for s in $(seq 1 $procs); do
  ( /tmp/fuzz_inflate $per $((1000 + s)) > /tmp/fuzz_inflate_$s.out 2> /tmp/fuzz_inflate_$s.err; echo "seed $((1000 + s)) exit $? : $(grep -E "DONE|PARALLEL|COMPRESSOR" /tmp/fuzz_inflate_$s.out | tr "\n" " ")" >> $log; [ -s /tmp/fuzz_inflate_$s.err ] && { echo "--- stderr of seed $((1000 + s))"; head -40 /tmp/fuzz_inflate_$s.err; } >> $log ) &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
cat $log
exit $rc
