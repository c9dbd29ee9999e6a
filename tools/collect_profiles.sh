#!/bin/bash
# Copies the summaries of tools/measure_round.sh <prefix> from gpurun_out/ (scratch) into profiles/ (tracked).
set -e
prefix=$1
for d in gpurun_out/prof_${prefix}_*; do
  [ -d "$d" ] || continue
  tag=$(basename "$d" | sed "s/^prof_//")          # <prefix>_<config>_<math>
  cp "$d/summary.txt" profiles/${tag}_summary.txt
  cp "$(ls "$d"/trace/*/*kernel_stats.csv | head -1)" profiles/${tag}_kernel_stats.csv
  cm=${tag#${prefix}_}                              # <config>_<math>
  [ -f "$d/traffic.json" ] && cp "$d/traffic.json" profiles/traffic_${cm}.json
  [ -f "$d/valu_busy.json" ] && cp "$d/valu_busy.json" profiles/valu_busy_${cm}.json
done
for f in section_profile.log section_single.md section_lanes.log section_hist.log; do
  [ -f gpurun_out/$f ] && cp gpurun_out/$f profiles/${prefix}_$f
done
[ -f gpurun_out/configs_${prefix}.md ] && cp gpurun_out/configs_${prefix}.md profiles/${prefix}_configs_1gpu.md
[ -f gpurun_out/bench_${prefix}.json ] && cp gpurun_out/bench_${prefix}.json profiles/${prefix}_bench_1gpu.json
ls profiles | grep "^${prefix}_"
