#!/bin/bash
# launch knobs re-swept on the round-3 kernel (C3 strict, bench command): rejection rounds per iteration, waves per CU
set -o pipefail
mkdir -p gpurun_out
log=gpurun_out/r3_sweep_knobs.log
: > $log
run() { # VAR value config
  r=$(env $1=$2 timeout -k 10 120 python3 bench.py --config $3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 0.8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f Gsamples/s %.4f ms_per_step' % (d['value']/1e9, d['ms_per_step']))") || exit 1
  echo "$3 $1=$2: $r" | tee -a $log
}
for v in 1 2 3 4 6; do run RT_REJECT_TRIES $v C3; done
for v in 1 2 3 4 6; do run RT_REJECT_TRIES $v C2; done
for v in 20 24 28 32; do run RT_WAVES_PER_CU $v C3; done
for v in 20 24 28 32; do run RT_WAVES_PER_CU $v C2; done
for v in 16 20 24 28; do run RT_WAVES_PER_CU $v C4; done
