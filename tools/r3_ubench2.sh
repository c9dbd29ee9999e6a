#!/bin/bash
# Round-3 microbenchmarks, second visit: per-SIMD spans in valu_ceiling, instruction classes + patterns (wall and the
# VALU2 dual-issue counter), MFMA co-execution with the MFMA waves older / prioritised, and the dual-issue share of the
# path kernel itself (C3 / C4 strict).
set -o pipefail
export TMPDIR=/tmp
out=$PWD/gpurun_out/r3u
mkdir -p $out
pushd tools/ubench > /dev/null
timeout -k 10 300 ./valu_ceiling > $out/valu_ceiling.txt 2>&1 || { tail -3 $out/valu_ceiling.txt; exit 1; }
timeout -k 10 300 ./valu_classes > $out/valu_classes.txt 2>&1 || { tail -3 $out/valu_classes.txt; exit 1; }
cat $out/valu_classes.txt
timeout -k 10 120 ./mfma_coexec > $out/mfma_coexec.txt 2>&1 || { tail -3 $out/mfma_coexec.txt; exit 1; }
cat $out/mfma_coexec.txt
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES --output-format csv -d $out/cl_a -- ./valu_classes pmc > $out/cl_a.log 2>&1 || { echo "classes pmc failed"; tail -3 $out/cl_a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/cl_c -- ./valu_classes pmc > $out/cl_c.log 2>&1 || { echo "classes pmc 2 failed"; exit 1; }
popd > /dev/null
for spec in "C3 strict" "C4 strict" "C2 strict"; do
  set -- $spec
  d=gpurun_out/prof_r3a_$1_$2; mkdir -p $d
  cmd="bench.py --steps 20 --warmup 20 --min-seconds 0 --no-cpu-baseline --no-extras --config $1 --math $2"
  timeout -s KILL 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS --output-format csv -d $d/pmc_VALU2 -- python3 $cmd > $d/pmc_VALU2.log 2>&1 || { echo "pass VALU2 $1 failed"; tail -3 $d/pmc_VALU2.log; exit 1; }
  timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $d/pmc_A -- python3 $cmd > $d/pmc_A.log 2>&1 || { echo "pass A $1 failed"; exit 1; }
  timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $d/pmc_G -- python3 $cmd > $d/pmc_G.log 2>&1 || { echo "pass G $1 failed"; exit 1; }
  python3 tools/prof_summary.py $d > $d/summary.txt 2>&1
  grep -A 14 "rt_path_kernel<true, false" $d/summary.txt | head -18
done
