#!/bin/bash
# C4 (4 spp: a lane needs a new pixel every 4 samples) against the frame group of its staged launches
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r3_c4_groups.log; : > $out
for g in 1 2 4 8 16; do
  for m in strict fast; do
    r=$(RT_FRAME_GROUP=$g timeout -k 10 120 python3 bench.py --config C4 --math $m --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 1.5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f Gsamples/s %.4f ms_per_step' % (d['value']/1e9, d['ms_per_step']))") || exit 1
    echo "RT_FRAME_GROUP=$g C4 $m: $r" | tee -a $out
  done
done
r=$(timeout -k 10 120 python3 bench.py --config C4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 1.5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f Gsamples/s %.4f ms_per_step' % (d['value']/1e9, d['ms_per_step']))")
echo "default C4 strict: $r" | tee -a $out
