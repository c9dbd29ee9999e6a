#!/bin/bash
# The measurements a round's documents quote, in one GPU visit: rocprofv3 trace + PMC of the bench command per
# config and arithmetic mode, the section profiles (when their build_variants/ libraries were built beforehand),
# the five-config table, the contract bench line. usage: tools/measure_round.sh <prefix> [prof|sections|configs|bench ...]
# Results land in gpurun_out/; tools/collect_profiles.sh <prefix> copies what is judged into profiles/.
set -o pipefail
prefix=$1; shift
what=${*:-prof sections configs bench}
mkdir -p gpurun_out
for w in $what; do
  case $w in
    prof)
      for spec in "C3 strict" "C3 fast" "C4 strict" "C4 fast" "C2 strict"; do
        set -- $spec
        echo "== prof $1 $2"
        bash tools/prof.sh ${prefix}_$1_$2 --config $1 --math $2 > gpurun_out/prof_${prefix}_$1_$2.log 2>&1 || { tail -5 gpurun_out/prof_${prefix}_$1_$2.log; exit 1; }
        grep -E "rt_path_kernel<(true|false), false" gpurun_out/prof_${prefix}_$1_$2/summary.txt | head -1 | cut -c1-220
      done ;;
    sections)
      bash tools/section_profile.sh > gpurun_out/sp_run.log 2>&1 || { tail -5 gpurun_out/sp_run.log; exit 1; }
      tail -12 gpurun_out/section_single.md 2>/dev/null ;;
    configs)
      timeout -k 10 600 python3 -u tools/run_configs.py > gpurun_out/configs_${prefix}.md 2> gpurun_out/configs_${prefix}.err || { tail -5 gpurun_out/configs_${prefix}.err; exit 1; }
      cat gpurun_out/configs_${prefix}.md ;;
    bench)
      timeout -k 10 120 python3 -u __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; exit 1; }
      tail -1 gpurun_out/smoke.log
      timeout -k 10 400 python3 -u bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${prefix}.json 2> gpurun_out/bench_${prefix}.err || { tail -5 gpurun_out/bench_${prefix}.err; exit 1; }
      cut -c1-600 gpurun_out/bench_${prefix}.json ;;
  esac
done
