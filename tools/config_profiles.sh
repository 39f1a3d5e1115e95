#!/bin/bash
# rocprofv3 kernel stats + the bench line (with its kernel_table) for BASELINE config 3 (CHAMMI, variable C) and config 5 (Base, 64 channels,
# N = 12 545).  usage (on the GPU box): bash tools/config_profiles.sh OUTDIR
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/${1:-r05_cfg}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift
  (cd $ROOT && timeout -k 10 400 python3 bench.py --no-cpu-baseline "$@" > $OUT/${name}_bench.json 2> $OUT/${name}_bench.err); tail -1 $OUT/${name}_bench.json | cut -c1-160
  rm -rf /tmp/cfgstats_$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cfgstats_$name -o s -- python3 $ROOT/bench.py --no-graph --no-cpu-baseline "$@" > $OUT/${name}_stats.log 2>&1
  find /tmp/cfgstats_$name -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \;
}
run chammi --chammi --steps 10 --warmup 3
run hcs --hcs --steps 10 --warmup 3
run base64ch --arch base --channels 64 --batch 2 --steps 4 --warmup 2
ls -la $OUT
