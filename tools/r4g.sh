#!/bin/bash
set -o pipefail
out=gpurun_out/r4g; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -s -k "loss_curve" 2>&1 | grep -E "loss-curve|passed|failed|Error|assert" | tee $out/loss_curves.txt
GB_ZERO=1 GB_ROUNDS=8 timeout -k 10 240 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee $out/gemm_bench_pair_zero.txt
