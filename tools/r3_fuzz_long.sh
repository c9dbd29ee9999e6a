#!/bin/bash
# A longer soundness sweep of the shipped library (see tools/r3_fuzz_final.sh), other seeds
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r3_fuzz_long.log; : > $out
run() { echo "== fuzz_proofs.py $*" >> $out; timeout -k 10 1000 python3 -u tools/fuzz_proofs.py "$@" 2>&1 | tee -a gpurun_out/r3_fuzz_long_progress.log | grep -v "^\.\.\." >> $out || { tail -3 $out; exit 1; }; tail -1 $out; }
if [ -z "$FUZZ_FAST_ONLY" ]; then
run 1000000 200000
run 2000000 60000 --big
fi
run 3000000 40000 --fast
run 4000000 15000 --big --fast
