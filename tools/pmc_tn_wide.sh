#!/bin/bash
# fc2's weight gradient (P 384, Q 1536) on the shipped 384 x 128 tile and on the 384 x 256 variant (libdcv_hip_tnw.so, -DDCV_TN_WIDE_Q=1): L1 request counters of both.
ROOT=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
out=$ROOT/gpurun_out/pmc_tn_wide; rm -rf $out; mkdir -p $out
for v in base tnw; do
  if [ $v = base ]; then export DCV_LIB=$ROOT/diverse_channel_vit_amd/libdcv_hip.so; else export DCV_LIB=$ROOT/diverse_channel_vit_amd/libdcv_hip_$v.so; fi
  i=0
  for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
             "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/ptw_$v/p$i -o p -- python3 $ROOT/tools/tn_one.py > $out/$v.p$i.log 2>&1 || { echo "$v pass $i failed"; tail -3 $out/$v.p$i.log; }
  done
  python3 $ROOT/tools/pmc_read.py /tmp/ptw_$v gemm_tn384_group > $out/$v.txt 2>&1
  grep "max rel err" $out/$v.p1.log
done
echo done
