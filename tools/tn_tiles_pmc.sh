#!/bin/bash
# HBM traffic (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes; FETCH x 2 on gfx950) of the weight-gradient kernels per tile shape and
# reduction mode: tools/tn_tiles_bench.py under the profiler.  Output: gpurun_out/<dir>/tn_tiles_pmc.txt
set -e
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/${1:-tn_tiles_pmc}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "/tmp/tntiles_${c}"
  TB_ROUNDS=3 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/tntiles_$c -o p -- python3 $ROOT/tools/tn_tiles_bench.py > $OUT/pmc_$c.log 2>&1 && echo "$c done"
done
python3 - <<'PY' > $OUT/tn_tiles_pmc.txt
import csv, glob, collections, re
# launch order of tools/tn_tiles_bench.py with TB_ROUNDS=3: per shape, 3 rounds x cases x 3 launches (legal tiles x (atomic, det))
shapes = [(1536, 384), (384, 1536), (1152, 384), (384, 384)]
def cases(P, Q):
    out = []
    for n in ("narrow", "wide"):
        if n == "wide" and (P % 384 or Q % 128): continue
        out += [(n, "atomic"), (n, "det")]
    return out
order = [(P, Q, n, mode) for (P, Q) in shapes for rnd in range(3) for (n, mode) in cases(P, Q) for rep in range(3)]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
red = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(f"/tmp/tntiles_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == c and re.search(r"gemm_tn\w*_kernel|det_reduce", row["Kernel_Name"]):
                rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"])))
    rows.sort()
    i = -1
    for d, k, v in rows:
        if "det_reduce" in k:
            if i >= 0: red[order[i]][c].append(v)
            continue
        i += 1
        if i < len(order): agg[order[i]][c].append(v)
print("# HBM MB per launch (rocprofv3 --pmc: 2 x FETCH_SIZE + WRITE_SIZE; gfx950 correction), mean over 9 launches; det = kernel + its reduction pass")
seen = []
for key in order:
    if key in seen: continue
    seen.append(key)
    d = agg[key]; r = red[key]
    rd = 2 * 1024 * sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1) / 1e6
    wr = 1024 * sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1) / 1e6
    rrd = 2 * 1024 * sum(r["FETCH_SIZE"]) / max(len(r["FETCH_SIZE"]), 1) / 1e6 if r["FETCH_SIZE"] else 0.0
    rwr = 1024 * sum(r["WRITE_SIZE"]) / max(len(r["WRITE_SIZE"]), 1) / 1e6 if r["WRITE_SIZE"] else 0.0
    P, Q, n, mode = key
    alg = 2.0 * 100416 * (P + Q) / 1e6
    print(f"P{P:5d} Q{Q:5d} {n:7s} {mode:7s}: kernel read {rd:7.1f} written {wr:6.1f} | reduction read {rrd:6.1f} written {rwr:5.1f} | total {rd + wr + rrd + rwr:7.1f} MB  (operands once: {alg:6.1f} MB)")
PY
cat $OUT/tn_tiles_pmc.txt
