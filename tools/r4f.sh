#!/bin/bash
set -o pipefail
out=gpurun_out/r4f; mkdir -p $out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; rc=$?
tail -15 $out/pytest_gpu.log
grep -E "loss-curve|attention at|fused vs pair" $out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || { tail $out/bench.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["step_roofline"], d["cpu_baseline"])
PY
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --hcs --no-cpu-baseline > $out/bench_hcs.json 2> $out/bench_hcs.err || { tail $out/bench_hcs.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$out/bench_hcs.json").read().strip().splitlines()[-1])
print("hcs", d["value"], d["ms_per_step"], d["config"]["host_syncs_per_step"], d["config"]["channels_per_step"], d["config"]["tokens_per_sec"])
PY
DCV_HCS_ON_DEVICE=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --hcs --no-cpu-baseline > $out/bench_hcs_legacy.json 2> $out/bench_hcs_legacy.err
python - <<PY
import json
d=json.loads(open("$out/bench_hcs_legacy.json").read().strip().splitlines()[-1])
print("hcs legacy", d["value"], d["ms_per_step"], d["config"]["host_syncs_per_step"], d["config"]["channels_per_step"])
PY
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --graph --no-cpu-baseline > $out/bench_graph.json 2> $out/bench_graph.err
python - <<PY
import json
d=json.loads(open("$out/bench_graph.json").read().strip().splitlines()[-1])
print("graph", d["value"], d["ms_per_step"])
PY
timeout -k 10 200 python bench.py --steps 10 --warmup 4 --force-dp --no-cpu-baseline > $out/bench_forcedp.json 2> $out/bench_forcedp.err || tail -5 $out/bench_forcedp.err
python - <<PY
import json
d=json.loads(open("$out/bench_forcedp.json").read().strip().splitlines()[-1])
print("force-dp", d["value"], d["ms_per_step"], d["dp"])
PY
for ws in 0 1; do for gr in "" "--graph"; do
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --wgrad-stream $ws $gr --no-cpu-baseline > $out/bench_ws${ws}${gr}.json 2> $out/bench_ws${ws}${gr}.err
python - <<PY
import json
d=json.loads(open("$out/bench_ws${ws}${gr}.json").read().strip().splitlines()[-1])
print("wgrad-stream $ws $gr", d["value"], d["ms_per_step"], d["median_ms_per_step"])
PY
done; done
GB_ROUNDS=8 timeout -k 10 240 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee $out/gemm_bench_pair.txt
