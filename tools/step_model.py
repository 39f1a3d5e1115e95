"""The three-term step model of DESIGN.md section 6 applied kernel by kernel: predicted = 0.78 us per GFLOP executed on the matrix pipe
+ 0.16 us per MB of HBM traffic (PMC) + 0.033 us per MB moved L2 -> LDS by the operand DMA (computed from the tile shapes), against the
measured one-stream launch time.  Input: a bench.py JSON line (its kernel_table).   python tools/step_model.py profiles/r03_v2_bench.json"""
import json, math, re, sys

A_MFMA, A_HBM, A_DMA = 0.78, 0.16, 0.033  # us per GFLOP, per MB, per MB


def dims(shape):
    return {k: int(v) for k, v in re.findall(r"([A-Za-z]+)(\d+)", shape)}


def dma_mb(sym, shape):
    d = dims(shape)
    if sym.startswith("gemm_nt384"):
        tiles = math.ceil(d["M"] / 256) * (d["N"] // 384)
        return tiles * (256 + 384) * d["K"] * 2 / 1e6
    if sym.startswith("gemm_nt_kernel"):
        tiles = math.ceil(d["M"] / 256) * math.ceil(d["N"] / 128)
        return tiles * (256 + 128) * d["K"] * 2 / 1e6
    if sym.startswith("gemm_tn384"):
        return (d["P"] // 384) * (d["Q"] // 128) * (384 + 128) * d["M"] * 2 / 1e6
    if sym.startswith("gemm_tn_kernel"):
        return math.ceil(d["P"] / 128) * math.ceil(d["Q"] / 128) * 256 * d["M"] * 2 / 1e6  # register-staged, same L2 -> CU bytes
    if sym.startswith("attn_"):
        wgs = d["B"] * d["H"] * math.ceil((d["Nq"] if "dkdv" not in sym else d["N"]) / 128)
        rows = d["N"] if "dkdv" not in sym else d["Nq"]
        return wgs * rows * 2 * 64 * 2 / 1e6
    return 0.0


line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rows, tot_p, tot_m = [], 0.0, 0.0
print(f"{'symbol':26s} {'shape':24s} {'n':>4s} {'GFLOP':>7s} {'HBM MB':>7s} {'DMA MB':>7s} | {'mfma':>6s} {'hbm':>6s} {'dma':>6s} {'model':>7s} {'meas.':>7s} {'ratio':>6s}")
for r in line["kernel_table"]:
    gf = r.get("gflop_executed", 0.0)
    hbm = r.get("pmc_mbytes_per_launch") or r.get("algorithmic_mbytes", 0.0)
    if r["shape"].startswith("M64 ") or "Nq1" in r["shape"].split()[-1:][0] and r["shape"].endswith("Nq1"):
        hbm = r.get("algorithmic_mbytes", hbm)  # the PMC mean of a symbol is over its large launches
    dma = dma_mb(r["symbol"], r["shape"])
    t = (A_MFMA * gf, A_HBM * hbm, A_DMA * dma)
    pred = sum(t)
    n = r["launches_per_step"]
    tot_p += pred * n
    tot_m += r["avg_us"] * n
    print(f"{r['symbol']:26s} {r['shape']:24s} {n:4.0f} {gf:7.1f} {hbm:7.1f} {dma:7.0f} | {t[0]:6.1f} {t[1]:6.1f} {t[2]:6.1f} {pred:7.1f} {r['avg_us']:7.1f} {pred / r['avg_us']:6.2f}")
print(f"listed kernels: model {tot_p / 1e3:.2f} ms per step, measured {tot_m / 1e3:.2f} ms (one-stream launch times; the step itself: {line['ms_per_step']:.2f} ms on two streams)")
