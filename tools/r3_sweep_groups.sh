#!/bin/bash
# staged launches: frames per unit of work (RT_FRAME_GROUP, rt_api.cpp) on C2 / C4 (20-pass renders) and on rank 3 of 8 of C3
set -o pipefail
mkdir -p gpurun_out
log=gpurun_out/r3_sweep_groups.log
: > $log
run() { # VAR value config
  r=$(env $1=$2 $4 timeout -k 10 120 python3 bench.py --config $3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 1.0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f Gsamples/s %.4f ms_per_step' % (d['value']/1e9, d['ms_per_step']))") || exit 1
  echo "$3 $1=$2 $4: $r" | tee -a $log
}
for c in C2 C4; do for v in 1 2 4 8 16; do run RT_FRAME_GROUP $v $c; done; done
