mkdir -p gpurun_out/r4af
for i in 1 2 3; do
  for ps in 0 1; do
    DCV_ATTN_PS=$ps timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ps=$ps', d['value'], d['ms_per_step'], d['median_ms_per_step'])" >> gpurun_out/r4af/ab_ps.txt
  done
done
cat gpurun_out/r4af/ab_ps.txt
