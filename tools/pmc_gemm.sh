#!/bin/bash
# separate --pmc passes over the GEMM kernels at the headline shapes (tools/gemm_only.py); result databases land in gpurun_out/pmc_gemm/p<i>/
# (counters only with --kernel-trace: no other trace domain beside --pmc on this pool)
set -e
ROOT=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/pmc_gemm; mkdir -p $ROOT/gpurun_out/pmc_gemm
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp -d $ROOT/gpurun_out/pmc_gemm/p$i -o p -- python3 $ROOT/tools/gemm_only.py 3 > $ROOT/gpurun_out/pmc_gemm/p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
python3 $ROOT/tools/pmc_gemm_read.py $ROOT/gpurun_out/pmc_gemm > $ROOT/gpurun_out/pmc_gemm/summary.txt
cat $ROOT/gpurun_out/pmc_gemm/summary.txt
