#!/bin/bash
for b in 1 2 4 8 16; do
  echo "== RT_BATCH_LANES=$b"
  RT_BATCH_LANES=$b timeout -k 10 120 python -u tools/quick_perf.py c2 c3 || exit 1
done
