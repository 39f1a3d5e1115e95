#!/bin/bash
# the secondary bench lines of DESIGN.md section 6 on one box: HCS, CHAMMI, --graph, --h2d, forced DP collectives, Base 64 ch
mkdir -p gpurun_out/sec; rm -f gpurun_out/sec/runs.txt
run() { name=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['unit'], d['ms_per_step'], {k: d['config'][k] for k in ('tokens_per_sec','host_syncs_per_step','channels_per_step') if k in d['config']})" >> gpurun_out/sec/runs.txt; }
run headline --steps 30 --warmup 8
run hcs --hcs --steps 30 --warmup 8
run chammi --chammi --steps 30 --warmup 8
run graph --graph --steps 30 --warmup 8
run h2d --h2d --steps 30 --warmup 8
run force_dp --force-dp --steps 20 --warmup 5
run base64ch --arch base --channels 64 --batch 2 --steps 6 --warmup 2
cat gpurun_out/sec/runs.txt
