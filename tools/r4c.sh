#!/bin/bash
set -o pipefail
out=gpurun_out/r4c; mkdir -p $out
L=diverse_channel_vit_amd
timeout -k 10 300 python tools/attn_bench.py $L/libdcv_hip_fa1.so $L/libdcv_hip_fa2.so 2>&1 | grep -v amdgpu.ids | tee $out/attn_bench_barrier_ablation.txt
