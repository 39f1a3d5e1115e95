#!/bin/bash
# round-end evidence: (1) default bench line, (2) rocprofv3 --kernel-trace --stats of a short eager bench, (3) separate --pmc
# passes for HBM traffic (FETCH_SIZE, WRITE_SIZE).  Small summaries under gpurun_out/v5/ (raw traces are deleted on the box).
set -e
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/${1:-r02_v1}; mkdir -p $OUT
cd $ROOT && timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err && tail -1 $OUT/bench.json | cut -c1-200
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/dcvstats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dcvstats -o s -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline > $OUT/stats.log 2>&1 && echo stats done
find /tmp/dcvstats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "/tmp/dcvpmc_${c}"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/dcvpmc_$c -o p -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $OUT/pmc_$c.log 2>&1 && echo "$c done"
done
DCV_OUT=${1:-r02_v1} python3 - <<'PY'
import csv, glob, json, collections, re, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/" + os.environ.get("DCV_OUT", "r02_v1") + "/pmc_traffic.json"
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/dcvpmc_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != c:
                continue
            k = row["Kernel_Name"]
            m = re.findall(r"(\w+_kernel)(<[^>]*>)?", k)
            name = (m[0][0] + m[0][1]) if m else k[:50]
            a = agg[name][c]; a[0] += float(row["Counter_Value"]); a[1] += 1
res = {}
for k, d in agg.items():
    n = max(d["FETCH_SIZE"][1], d["WRITE_SIZE"][1], 1)
    # counters are KiB; gfx950 reports half the bytes of wide coalesced reads: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section)
    rd = 2 * 1024 * d["FETCH_SIZE"][0] / max(d["FETCH_SIZE"][1], 1)
    wr = 1024 * d["WRITE_SIZE"][0] / max(d["WRITE_SIZE"][1], 1)
    res[k] = dict(launches=n, read_bytes_per_launch=rd, written_bytes_per_launch=wr)
res["_meta"] = dict(steps=3, command="bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline", counters="FETCH_SIZE x2 (gfx950) + WRITE_SIZE, KiB -> bytes")
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print("kernels:", len(res))
PY
ls -la $OUT
