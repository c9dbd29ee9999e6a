#!/bin/bash
# One pass per launch at 1 spp (the interactive case: a host that presents every frame) against the units a wave takes per atomic on the
# queue head (RT_QUEUE_CHUNK; unset = the rule of rt_api.cpp). bench.py's per_pass_launches leg.
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r3_queue_chunk.log; : > $out
for c in C5 C3; do
  for k in 1 2 4 8 16 auto; do
    if [ $k = auto ]; then unset RT_QUEUE_CHUNK; else export RT_QUEUE_CHUNK=$k; fi
    r=$(timeout -k 10 200 python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --min-seconds 1.0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); p=d['per_pass_launches']; print('batched %.2f Gsamples/s %.4f ms per pass; one launch per pass %.2f Gsamples/s %.4f ms' % (d['value']/1e9, d['ms_per_step'], p['value']/1e9, p['ms_per_step']))") || exit 1
    echo "$c RT_QUEUE_CHUNK=$k: $r" | tee -a $out
  done
done
