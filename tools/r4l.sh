#!/bin/bash
out=gpurun_out/r4l; mkdir -p $out
python __graft_entry__.py smoke 2>&1 | tail -2
timeout -k 10 200 python bench.py --steps 10 --warmup 4 --force-dp --no-cpu-baseline > $out/forcedp.json 2> $out/forcedp.err; echo "rc=$?"
python - <<PY
import json
d=json.loads(open("$out/forcedp.json").read().strip().splitlines()[-1])
print("force-dp", d["value"], d["ms_per_step"], json.dumps(d["dp"]["modes"]), d["dp"]["overlap"])
PY
DCV_BENCH_WATCHDOG_S=0.05 timeout -k 10 200 python bench.py --steps 10 --warmup 4 --force-dp --no-cpu-baseline > $out/forcedp_wd.json 2> $out/forcedp_wd.err; echo "rc=$?"
python - <<PY
import json
d=json.loads(open("$out/forcedp_wd.json").read().strip().splitlines()[-1])
print("force-dp watchdog", d["value"], d["ms_per_step"], json.dumps(d["dp"]["modes"]))
PY
