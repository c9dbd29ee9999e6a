#!/bin/bash
# Round-3 microbenchmarks, third visit: which slow instructions pair with a fast one (class S) and which issue alone (class X)
set -o pipefail
export TMPDIR=/tmp
out=$PWD/gpurun_out/r3u
mkdir -p $out
cd tools/ubench
timeout -k 10 400 ./valu_pairs > $out/valu_pairs.txt 2>&1 || { tail -3 $out/valu_pairs.txt; exit 1; }
cat $out/valu_pairs.txt
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 --output-format csv -d $out/pr_a -- ./valu_pairs pmc > $out/pr_a.log 2>&1 || { echo "pairs pmc failed"; tail -3 $out/pr_a.log; exit 1; }
