"""Adds the attention kernels' SQ counters (tools/pmc_attn.sh ... ps -> tools/pmc_read.py text) to the GEMM counter JSON of tools/pmc_gemm_read.py, under the
keys bench.py's kernel table uses ("symbol|shape"): python tools/pmc_merge_attn.py gemm_sq_counters.json pmc_attn.txt out.json"""
import collections, json, sys
src, txt, dst = sys.argv[1:4]
out = json.load(open(src))
res, cur = collections.OrderedDict(), None
for l in open(txt):
    if not l.startswith(" "):
        cur = l.strip(); res[cur] = {}
    elif cur:
        k, v = l.split(); res[cur][k] = float(v)
shape = "B64 N1569 H6 Nq1569"
for k, d in res.items():
    if "attn" not in k or "SQ_VALU_MFMA_BUSY_CYCLES" not in d:
        continue
    if k.endswith("<false>"):  # the plain entries: not on the step's path
        continue
    gui = d["GRBM_GUI_ACTIVE"] / 8.0
    busy = d["SQ_VALU_MFMA_BUSY_CYCLES"]
    sym = k if not k.startswith("attn_bwd_dkdv2") else "attn_bwd_dkdv2_kernel<true> (key remainder, TAIL2)"
    out[f"{sym}|{shape}"] = {"case": "attention, pre-scaled q, B64 H6 N1569 (tools/attn_only.py)", "avg_us_profiled": d["_dur_us"],
                            "mfma_busy_frac": round(busy / (gui * 1024), 4), "valu_mfma_coexec_frac": round(d["SQ_VALU_MFMA_COEXEC_CYCLES"] / busy, 4),
                            "valu_busy_frac": round(d.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (gui * 1024), 4),
                            "lds_bank_conflict_cycles": d.get("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles": d.get("SQ_LDS_IDX_ACTIVE"),
                            **{c: v for c, v in d.items() if not c.startswith("_")}}
json.dump(out, open(dst, "w"), indent=1)
print("\n".join(f"{k:70s} {v['avg_us_profiled']:8.1f} us  mfma_busy {v['mfma_busy_frac']:.3f}  coexec {v['valu_mfma_coexec_frac']:.3f}" for k, v in out.items()))
