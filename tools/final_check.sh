#!/bin/bash
# full GPU suite + default bench line of the tree as it stands (run at the end of a round: gpurun -- 'bash tools/final_check.sh')
mkdir -p gpurun_out/final
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/final/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/final/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 400 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; echo "bench rc=$?"; cut -c1-260 gpurun_out/final/bench.json | tail -1
