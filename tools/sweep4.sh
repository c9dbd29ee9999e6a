#!/bin/bash
for o in 0 1; do
  echo "== RT_TILE_ORDER=$o"
  RT_TILE_ORDER=$o timeout -k 10 200 python -u tools/quick_perf.py c2 c3 c3d c4 || exit 1
done
