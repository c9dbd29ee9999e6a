#!/bin/bash
# sweep tuning knobs with the quick perf probe (fast + strict)
for t in 1 2 3 4 64; do
  echo "== RT_REJECT_TRIES=$t"
  RT_REJECT_TRIES=$t timeout -k 10 120 python -u tools/quick_perf.py c2 c3 c4 || exit 1
done
for w in 8 16 24 32; do
  echo "== RT_WAVES_PER_CU=$w"
  RT_WAVES_PER_CU=$w timeout -k 10 120 python -u tools/quick_perf.py c3 c3d || exit 1
done
