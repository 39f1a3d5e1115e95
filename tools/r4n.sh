#!/bin/bash
mkdir -p gpurun_out/r4n
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r4n/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4n/pytest_gpu.log
bash tools/pmc_traffic.sh r04_v2 2>&1 | tail -6
