#!/bin/bash
# builds the -DDCV_PAIR_OFFSET variants (all with stamps) for tools/gemm_pair_phase.py
cd "$(dirname "$0")/.."
python - <<'PY'
from diverse_channel_vit_amd._build import build_variant
for off in (0, 6000, 12000, 18000):
    build_variant("pair%d" % off, ["DCV_PAIR_OFFSET=%d" % off, "DCV_PAIR_STAMP=1"])
PY
