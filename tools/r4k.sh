#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -q -s -k "loss_curve" 2>&1 | grep -E "loss-curve|passed|failed|assert" 
DCV_FUSE_LN=0 timeout -k 10 600 python -m pytest tests/test_model_gpu.py -q -s -k "distinct" 2>&1 | grep -E "loss-curve|passed|failed|assert" 
