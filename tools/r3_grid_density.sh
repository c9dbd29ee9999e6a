#!/bin/bash
# C4 (1024 spheres) against the grid's density (cells per sphere; RT_GRID_DENSITY), and without the grid
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r3_grid_density.log
: > $out
for d in 0 0.5 1 2 4 8 16 32; do
  if [ $d = 0 ]; then export RT_GRID=0; echo "== RT_GRID=0" >> $out; else unset RT_GRID; export RT_GRID_DENSITY=$d; echo "== RT_GRID_DENSITY=$d" >> $out; fi
  RT_GRID_VERBOSE=1 QP_REPS=16 timeout -k 10 120 python3 -u tools/quick_perf.py c4 2>&1 | grep -E "C4|grid" | sort -u >> $out || exit 1
done
cat $out
