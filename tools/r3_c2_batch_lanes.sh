#!/bin/bash
# C2 (and C5) against the number of parked lanes that trigger the service section (RT_BATCH_LANES), current kernel
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r3_c2_batch_lanes.log; : > $out
for c in C2 C5; do
for b in 2 3 4 5 6 8 10; do
  r=$(RT_BATCH_LANES=$b timeout -k 10 120 python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 1.5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f Gsamples/s %.4f ms_per_step' % (d['value']/1e9, d['ms_per_step']))") || exit 1
  echo "RT_BATCH_LANES=$b $c strict: $r" | tee -a $out
done
done
