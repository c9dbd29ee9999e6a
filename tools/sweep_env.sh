#!/bin/bash
# Sweeps the launch knobs the library reads from the environment (scratch tool): usage tools/sweep_env.sh VAR v1 v2 ... -- configs
var=$1; shift
vals=()
while [ "$1" != "--" ]; do vals+=("$1"); shift; done
shift
mkdir -p gpurun_out
for v in "${vals[@]}"; do
  echo "== $var=$v" | tee -a gpurun_out/sweep_env.log
  env $var=$v timeout -k 10 200 python3 -u tools/quick_perf.py "$@" 2>&1 | grep -E "strict|fast" | cut -c1-100 | tee -a gpurun_out/sweep_env.log
done
