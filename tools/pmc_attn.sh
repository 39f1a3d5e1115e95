#!/bin/bash
# separate --pmc passes over the attention kernels; CSVs land in gpurun_out/pmc_attn/<pass>/
set -e
ROOT=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_SALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp -d $ROOT/gpurun_out/pmc_attn/p$i -o p -- python3 $ROOT/tools/attn_only.py 2 > $ROOT/gpurun_out/pmc_attn_p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
