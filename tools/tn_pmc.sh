#!/bin/bash
# HBM fetch of the weight-gradient GEMMs per library build: tools/tn_pmc.sh out_dir lib1.so [lib2.so ...]  (on the GPU box)
set -e
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$1; shift; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "/tmp/tnpmc_${name}_${c}"
    TB_ROUNDS=3 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/tnpmc_${name}_$c -o p -- python3 $ROOT/tools/tn_bench.py $ROOT/$lib > $OUT/pmc_${name}_$c.log 2>&1
  done
  python3 - "$name" >> $OUT/tn_pmc.txt <<'PY'
import csv, glob, sys, collections
name = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/tnpmc_{name}_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != c or "gemm_tn" not in row["Kernel_Name"]:
                continue
            k = "tn384" if "tn384" in row["Kernel_Name"] else "tn128"
            a = agg[k][c]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, d in agg.items():
    rd = 2 * 1024 * d["FETCH_SIZE"][0] / max(d["FETCH_SIZE"][1], 1)
    wr = 1024 * d["WRITE_SIZE"][0] / max(d["WRITE_SIZE"][1], 1)
    print(f"{name:24s} {k}: read {rd/1e6:7.1f} MB/launch  written {wr/1e6:6.1f} MB/launch  ({d['FETCH_SIZE'][1]} launches, mean over the shapes of tools/tn_bench.py)")
PY
done
cat $OUT/tn_pmc.txt
