#!/bin/bash
mkdir -p gpurun_out
bash tools/pmc_traffic.sh r04_v1 2>&1 | tail -15
timeout -k 10 300 python -m pytest tests/test_model_gpu.py -x -q -s -k "distinct_batches" 2>&1 | grep -E "loss-curve|passed|failed" 
