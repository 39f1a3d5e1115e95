#!/bin/bash
# separate --pmc passes over the attention kernels; usage: tools/pmc_attn.sh OUTNAME [DCV_LIB path] [what: all|pair|fused|ps]; summaries in gpurun_out/OUTNAME.txt
ROOT=$GRAFT_REPO_ROOT; name=${1:-pmc_attn}; lib=$2; what=${3:-all}
cd /tmp; export TMPDIR=/tmp
[ -n "$lib" ] && export DCV_LIB=$ROOT/$lib
rm -rf /tmp/$name
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp -d /tmp/$name/p$i -o p -- python3 $ROOT/tools/attn_only.py 2 $what > /tmp/$name.p$i.log 2>&1 || { echo "pass $i failed"; tail -5 /tmp/$name.p$i.log; }
done
python3 $ROOT/tools/pmc_read.py /tmp/$name attn > $ROOT/gpurun_out/$name.txt 2>&1
echo "$name done"
