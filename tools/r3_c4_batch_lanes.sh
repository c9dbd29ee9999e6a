#!/bin/bash
# C4 with the grid trace against the number of parked lanes that trigger the service section (RT_BATCH_LANES)
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r3_c4_batch_lanes.log; : > $out
for b in 2 3 4 6 8 12 16 24; do
  r=$(RT_BATCH_LANES=$b timeout -k 10 120 python3 bench.py --config C4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 1.5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f Gsamples/s %.4f ms_per_step' % (d['value']/1e9, d['ms_per_step']))") || exit 1
  echo "RT_BATCH_LANES=$b C4 strict: $r" | tee -a $out
done
