#!/bin/bash
# batch_lanes (parked lanes that trigger the frame-end / hand-out service) re-swept on the round-3 kernel
set -o pipefail
mkdir -p gpurun_out
log=gpurun_out/r3_sweep_batch_lanes.log
: > $log
for cfg in C2 C3 C5; do
  for v in 2 3 4 6 8 12 16 24; do
    r=$(RT_BATCH_LANES=$v timeout -k 10 120 python3 bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 0.6 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value']/1e9, d['ms_per_step'])") || exit 1
    echo "$cfg RT_BATCH_LANES=$v Gsamples/s ms_per_step: $r" | tee -a $log
  done
done
