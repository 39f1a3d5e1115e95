#!/bin/bash
# What bounds the NT GEMMs inside a CU: the vector L1's (TCP) requests to L2 — reads, writes, their summed latencies, the cycles the L1 sits on a full
# pending-request queue — in separate --pmc passes over tools/gemm_only.py (every GEMM entry at the headline shapes).
# usage: tools/pmc_gemm_tcp.sh ; summary gpurun_out/pmc_gemm_tcp/summary.txt  (counters only with --kernel-trace on this pool)
ROOT=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
out=$ROOT/gpurun_out/pmc_gemm_tcp; rm -rf $out; mkdir -p $out
i=0
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_READ_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp -d $out/p$i -o p -- python3 $ROOT/tools/gemm_only.py 3 > $out/p$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -3 $out/p$i.log; }
  echo "pass $i done"
done
python3 $ROOT/tools/pmc_gemm_read.py $out > $out/summary.txt 2> $out/summary.err
mv $out/gemm_sq_counters.json $out/gemm_tcp_counters.json
rm -rf $out/p*/   # the databases are large; the summary and the JSON are what travels back
echo "pmc_gemm_tcp done"
