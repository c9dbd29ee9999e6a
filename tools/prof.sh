#!/bin/bash
# rocprofv3 passes over the bench command. Usage: tools/prof.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
out=gpurun_out/prof_$tag; rm -rf $out  # (a directory left by an earlier run would be mixed into the summary)
mkdir -p $out
export TMPDIR=/tmp
# 20 + 20 passes: the library runs them as two 20-pass launches (it batches up to 64 recorded passes, 32 at 4K), the
# launch shape of the driver's bench command (--steps 20); one repetition, no CPU leg, no side measurements
cmd="bench.py --steps 20 --warmup 20 --min-seconds 0 --no-cpu-baseline --no-extras $*"
export PROF_FRAMES_PER_LAUNCH=20
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $cmd > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS" \
            "SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS" \
            "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
            "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d $out/pmc_$name -- python3 $cmd > $out/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/pmc_$name.log; }
done
python3 tools/prof_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
