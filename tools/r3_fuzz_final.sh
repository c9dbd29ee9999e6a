#!/bin/bash
# Soundness sweep of every test-skipping proof on the shipped library (tile masks incl. the occlusion drop, candidate lists, own-sphere
# drop, walk deferral, candidate queue, grid trace forced on) against the trace of every sphere by every ray: tools/fuzz_proofs.py
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r3_fuzz_final.log; : > $out
run() { echo "== fuzz_proofs.py $*" >> $out; timeout -k 10 900 python3 -u tools/fuzz_proofs.py "$@" 2>&1 | grep -v "^\.\.\." >> $out || { tail -3 $out; exit 1; }; tail -1 $out; }
run 100000 60000
run 200000 20000 --big
run 300000 10000 --fast
run 400000 5000 --big --fast
