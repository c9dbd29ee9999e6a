#!/bin/bash
# Round-3 microbenchmarks on the GPU box: VALU issue ceiling (+ PMC calibration) and MFMA / VALU co-execution.
set -o pipefail
export TMPDIR=/tmp
out=$PWD/gpurun_out/r3u
mkdir -p $out
cd tools/ubench
timeout -k 10 240 ./valu_ceiling > $out/valu_ceiling.txt 2>&1 || { tail -3 $out/valu_ceiling.txt; exit 1; }
tail -4 $out/valu_ceiling.txt
timeout -k 10 120 ./mfma_coexec > $out/mfma_coexec.txt 2>&1 || { tail -3 $out/mfma_coexec.txt; exit 1; }
cat $out/mfma_coexec.txt
timeout -k 10 60 rocprofv3 -L > $out/counters.txt 2>&1 || echo "counter list failed"
pass() {  # name binary args... ; counters in $PMC
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d $out/$name -- "$@" > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; return 1; }
}
PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES" pass vc_a ./valu_ceiling pmc &&
PMC="SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU2" pass vc_b ./valu_ceiling pmc &&
PMC="GRBM_GUI_ACTIVE GRBM_COUNT" pass vc_c ./valu_ceiling pmc &&
PMC="SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VALU" pass mc_a ./mfma_coexec &&
PMC="GRBM_GUI_ACTIVE GRBM_COUNT" pass mc_c ./mfma_coexec
ls $out
