"""Prints per-kernel mean counter values (and mean duration) from the rocprofv3 --pmc result databases of several passes."""
import sqlite3, glob, sys, collections
root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
res = collections.defaultdict(dict)
for f in sorted(glob.glob(root + "/*/*_results.db")):
    db = sqlite3.connect(f)
    for k, c, v, d, n, vg, lds in db.execute("select kernel_name, counter_name, avg(value), avg(duration), count(*), max(vgpr_count), max(lds_block_size) from counters_collection group by kernel_name, counter_name"):
        if filt in k:
            import re; kk = (re.findall(r"(\w+_kernel\w*(?:<[^>]*>)?)", k) or [k[:60]])[0]  # with the template arguments (plain / pre-scaled-q forms)
            res[kk][c] = v; res[kk]["_dur_us"] = d / 1e3; res[kk]["_n"] = n; res[kk]["_vgpr"] = vg; res[kk]["_lds"] = lds
for k, d in res.items():
    print(k)
    for c in sorted(d):
        print(f"    {c:32s} {d[c]:16.1f}")
