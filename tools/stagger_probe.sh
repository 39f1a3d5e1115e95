#!/bin/bash
# experiment: is the NT epilogue a chip-level write burst?  per-tile time vs number of active CUs, and with every other CU de-phased
for g in 256 128 64 32; do echo "grid $g"; DCV_NT_GRID=$g python tools/ab_bench.py diverse_channel_vit_amd/libdcv_hip.so 2>&1 | grep "^1 " | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l[l.index('{'):]); print({k:round(v,1) for k,v in d.items() if 'gemm' in k and 'tn' not in k})"; done
# (the de-phasing arm of this probe — every other CU sleeping 2..12 x 1024 clocks before its first tile — measured +-1 % and its kernel knob was removed)
