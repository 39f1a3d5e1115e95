#!/bin/bash
# usage: tools/gemm_variants.sh OUTDIR name1 name2 ...   -> runs tools/gemm_bench.py once per variant library (libdcv_hip_<name>.so; "base" = the product library)
out=$1; shift
mkdir -p $out
for v in "$@"; do
  if [ "$v" = base ]; then lib=diverse_channel_vit_amd/libdcv_hip.so; else lib=diverse_channel_vit_amd/libdcv_hip_$v.so; fi
  echo "== $v" | tee -a $out/gemm_variants.txt
  DCV_LIB=$PWD/$lib GB_ROUNDS=${GB_ROUNDS:-8} timeout -k 10 240 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $out/gemm_variants.txt || exit 1
done
