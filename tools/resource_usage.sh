#!/bin/bash
# Prints registers / LDS / occupancy of every kernel in one csrc file: tools/resource_usage.sh gemm.hip [extra flags]
f=$1; shift
extra=""
case $f in
  attn.hip) extra="-mllvm -amdgpu-mfma-vgpr-form=1";;
  attn_bwd.hip) extra="-mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize";;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast $extra "$@" -Rpass-analysis=kernel-resource-usage \
  -c "$(dirname "$0")/../diverse_channel_vit_amd/csrc/$f" -o /dev/null 2>&1 | \
  grep -E "Function Name|VGPRs:|AGPRs|SGPRs:|Occupancy|LDS Size|ScratchSize" | \
  sed -e 's/.*remark: [^ ]* *//' | paste - - - - - - - | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | sort -u
