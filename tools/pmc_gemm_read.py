"""Per GEMM case (tools/gemm_only.py's order: each case = REPS consecutive dispatches of one kernel) the mean counter values of the rocprofv3
--pmc passes under <dir>/p*/, the derived MFMA-busy fraction, and a JSON (<dir>/gemm_sq_counters.json) keyed by "symbol|shape" for bench.py."""
import collections, glob, json, os, re, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
root = sys.argv[1]
REPS = 3
names = ["nt qkv N1152 K384 bias", "nt fc1 N1536 K384 bias+gelu", "nt gelu-bwd N1536 K384", "nt dgrad N384 K1536 plain", "nt dgrad N384 K1152 plain",
         "nt dgrad N384 K384 plain", "nt fc2 N384 K1536 bias+resid", "nt proj N384 K384 bias+resid", "nt fc2 N384 K1536 bias+resid+LN", "nt proj N384 K384 bias+resid+LN", "tn wgrad P1152 Q384", "tn wgrad P1536 Q384",
         "tn wgrad P384 Q1536", "tn wgrad P384 Q384", "tn group fc2+fc1+proj+qkv"]
shapes = ["M100416 N1152 K384", "M100416 N1536 K384", "M100416 N1536 K384", "M100416 N384 K1536", "M100416 N384 K1152", "M100416 N384 K384",
          "M100416 N384 K1536", "M100416 N384 K384", "M100416 N384 K1536", "M100416 N384 K384", "M100416 P1152 Q384", "M100416 P1536 Q384", "M100416 P384 Q1536", "M100416 P384 Q384", "M100416 384x1536+1536x384+384x384+1152x384"]
res = collections.OrderedDict((n, {}) for n in names)
sym = {}
for f in sorted(glob.glob(root + "/p*/**/*_results.db", recursive=True)):
    db = sqlite3.connect(f)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    order = "dispatch_id" if "dispatch_id" in cols else ("start" if "start" in cols else "rowid")
    rows = list(db.execute(f"select {order}, kernel_name, counter_name, value, duration from counters_collection where kernel_name like '%gemm_%' order by {order}"))
    disp = collections.OrderedDict()
    for d, k, c, v, dur in rows:
        disp.setdefault(d, {"k": k, "dur": dur, "c": collections.defaultdict(float)})
        disp[d]["c"][c] += v
    dl = list(disp.values())
    if len(dl) != REPS * len(names):
        print(f"# {f}: {len(dl)} GEMM dispatches, expected {REPS * len(names)}: skipped", file=sys.stderr)
        continue
    for i, n in enumerate(names):
        grp = dl[REPS * i + 1:REPS * (i + 1)]  # the first launch of a case warms caches and clocks: dropped
        m = re.findall(r"(gemm_\w+_kernel)(<[^>]*>)?", grp[0]["k"])
        sym[n] = (m[0][0] + m[0][1]) if m else grp[0]["k"][:40]
        res[n]["_dur_us"] = sum(g["dur"] for g in grp) / len(grp) / 1e3
        for c in grp[0]["c"]:
            res[n][c] = sum(g["c"][c] for g in grp) / len(grp)
out = {}
print("# rocprofv3 --pmc, separate passes (tools/pmc_gemm.sh), per launch, mean of 2 launches after a warm-up launch; MI355X, 256 CUs x 4 SIMDs.")
print("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); coexec = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES;")
print("# clock = GRBM_GUI_ACTIVE / 8 / duration (reads high on launches this short: MI355X_MICROARCH.md, DVFS give-back)")
for (n, d), shp in zip(res.items(), shapes):
    if not d:
        continue
    gui = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    frac = busy / (gui * 1024) if gui else float("nan")
    co = d.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) / busy if busy else float("nan")
    clk = gui / (d["_dur_us"] * 1e-6) / 1e9 if gui else float("nan")
    print(f"{n:34s} {sym.get(n, ''):24s} {d['_dur_us']:7.1f} us  mfma_busy {frac:5.3f}  coexec {co:5.3f}  clock~{clk:4.2f} GHz  LDS bank conflict cycles {d.get('SQ_LDS_BANK_CONFLICT', float('nan')):12.0f} "
          f"of {d.get('SQ_LDS_IDX_ACTIVE', float('nan')):12.0f} LDS cycles")
    for c in sorted(k for k in d if not k.startswith("_")):
        print(f"      {c:30s} {d[c]:16.1f}")
    out[f"{sym.get(n, n)}|{shp}"] = {"case": n, "avg_us_profiled": d["_dur_us"], "mfma_busy_frac": round(frac, 4), "valu_mfma_coexec_frac": round(co, 4),
                                    "lds_bank_conflict_cycles": d.get("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles": d.get("SQ_LDS_IDX_ACTIVE"),
                                    **{k: v for k, v in d.items() if not k.startswith("_")}}
json.dump(out, open(os.path.join(root, "gemm_sq_counters.json"), "w"), indent=1)
