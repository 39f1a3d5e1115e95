#!/bin/bash
out=gpurun_out/r4m; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "resid_ln" 2>&1 | tail -3
for rep in 1 2; do for f in 1 0; do
DCV_FUSE_LN=$f timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_ln$f.json 2> $out/bench_ln$f.err || { tail $out/bench_ln$f.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$out/bench_ln$f.json").read().strip().splitlines()[-1])
print("fuse_ln=$f", d["value"], d["ms_per_step"], d["median_ms_per_step"])
if $rep == 1:
  for r in d["kernel_table"]:
    if "<6>" in r["symbol"]:
        print("   ", r["symbol"], r["shape"], r["launches_per_step"], r["avg_us"], r["ms_per_step"])
PY
done; done
