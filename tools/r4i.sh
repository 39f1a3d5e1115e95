#!/bin/bash
set -o pipefail
out=gpurun_out/r4i; mkdir -p $out
echo skip kernels

timeout -k 10 900 python -m pytest tests/test_model_gpu.py -q -s > $out/pytest_m.log 2>&1 || { tail -40 $out/pytest_m.log; grep -E "loss-curve" $out/pytest_m.log; exit 1; }
tail -3 $out/pytest_m.log; grep -E "loss-curve" $out/pytest_m.log
for f in 1 0; do
DCV_FUSE_LN=$f timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_ln$f.json 2> $out/bench_ln$f.err || { tail $out/bench_ln$f.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$out/bench_ln$f.json").read().strip().splitlines()[-1])
print("fuse_ln=$f", d["value"], d["ms_per_step"])
for r in d["kernel_table"]:
    if "ln_fwd" in r["symbol"] or "<6>" in r["symbol"] or "<2>" in r["symbol"]:
        print("   ", r["symbol"], r["shape"], r["launches_per_step"], r["avg_us"], r["ms_per_step"])
PY
done
