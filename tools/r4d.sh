#!/bin/bash
mkdir -p gpurun_out
bash tools/pmc_attn.sh r4d_pmc_attn_fabl3 diverse_channel_vit_amd/libdcv_hip_fabl3.so
cat gpurun_out/r4d_pmc_attn_fabl3.txt
