#!/bin/bash
for t in 1 2 3 4; do
  echo "== RT_REJECT_TRIES=$t"
  RT_REJECT_TRIES=$t timeout -k 10 120 python -u tools/quick_perf.py c2 c3 || exit 1
done
