#!/bin/bash
# A/B of the SPLIT kernel variants against the one-trace-per-iteration ones (RT_SPLIT=0), same library. Scratch tool.
mkdir -p gpurun_out
for v in 1 0; do
  echo "== RT_SPLIT=$v" | tee -a gpurun_out/ab_split.log
  RT_SPLIT=$v timeout -k 10 200 python3 -u tools/quick_perf.py "$@" 2>&1 | grep -E "strict|fast" | cut -c1-150 | tee -a gpurun_out/ab_split.log
done
