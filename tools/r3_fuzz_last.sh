#!/bin/bash
# The last soundness sweep of the round, on the library as shipped (grid trace, queue-free hand-out of short units, staged low-spp batches)
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r3_fuzz_last.log; : > $out
run() { echo "== fuzz_proofs.py $*" >> $out; timeout -k 10 1100 python3 -u tools/fuzz_proofs.py "$@" 2>&1 | tee -a gpurun_out/r3_fuzz_last_progress.log | grep -v "^\.\.\." >> $out || { tail -3 $out; exit 1; }; tail -1 $out; }
run 10000000 400000
run 20000000 100000 --big
run 30000000 80000 --fast
