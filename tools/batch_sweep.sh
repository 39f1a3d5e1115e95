#!/bin/bash
# throughput of the headline model at several batch sizes (cache-residency probe): tools/batch_sweep.sh 64 32 16
for b in "$@"; do
  python bench.py --batch $b --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch', d['config']['global_batch'], d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['roofline']['kernel'], d['roofline']['avg_launch_ms'], 'ms/launch')"
done
