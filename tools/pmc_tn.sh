#!/bin/bash
# what does gemm_tn384_group_kernel wait on?  Separate --pmc passes (SQ wait breakdown, LDS, TCP / TCC) over the grouped weight-gradient
# launch at the headline shapes (tools/tn_group_bench.py).  usage: tools/pmc_tn.sh OUTNAME ; summary in gpurun_out/OUTNAME.txt
ROOT=$GRAFT_REPO_ROOT; name=${1:-pmc_tn}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/$name
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_TAG_STALL_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum TCC_NORMAL_EVICT_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  TB_ROUNDS=2 timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp -d /tmp/$name/p$i -o p -- python3 $ROOT/tools/tn_group_bench.py > /tmp/$name.p$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -3 /tmp/$name.p$i.log; }
done
python3 $ROOT/tools/pmc_read.py /tmp/$name gemm_tn > $ROOT/gpurun_out/$name.txt 2>&1
echo "$name done"
