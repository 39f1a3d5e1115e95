#!/bin/bash
set -o pipefail
out=gpurun_out/r4b; mkdir -p $out
L=diverse_channel_vit_amd
timeout -k 10 300 python tools/attn_bench.py $L/libdcv_hip.so $L/libdcv_hip_fabl1.so $L/libdcv_hip_fabl2.so $L/libdcv_hip_fabl3.so 2>&1 | grep -v amdgpu.ids | tee $out/attn_fused_ablation.txt
for z in "" 1; do
  for v in base gabl1 gabl2 gabl3; do
    if [ "$v" = base ]; then lib=$L/libdcv_hip.so; else lib=$L/libdcv_hip_$v.so; fi
    echo "== $v zero=${z:-0}" | tee -a $out/gemm_ablation_zero_ab.txt
    GB_ZERO=$z DCV_LIB=$PWD/$lib GB_ROUNDS=8 timeout -k 10 240 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $out/gemm_ablation_zero_ab.txt || exit 1
  done
  for t in wide narrow; do
    echo "== stamp tile=$t zero=${z:-0}" | tee -a $out/gemm_stamp_zero_ab.txt
    GB_ZERO=$z STAMP_TILE=$t DCV_LIB=$PWD/$L/libdcv_hip_stamp.so timeout -k 10 120 python tools/gemm_stamp.py 2>&1 | grep -v amdgpu.ids | tee -a $out/gemm_stamp_zero_ab.txt || exit 1
  done
done
for f in 1 0; do
  DCV_ATTN_BWD_FUSED=$f timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_fused$f.json 2> $out/bench_fused$f.err || { tail $out/bench_fused$f.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$out/bench_fused$f.json").read().strip().splitlines()[-1])
print("fused=$f", d["value"], d["ms_per_step"], d.get("step_roofline"))
PY
done
