#!/bin/bash
mkdir -p gpurun_out
bash tools/pmc_attn.sh r4e_pmc_attn_full
cat gpurun_out/r4e_pmc_attn_full.txt | grep -A28 attn_bwd_fused_kernel
