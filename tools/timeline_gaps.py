"""Reads a rocprofv3 --kernel-trace CSV of bench.py and prints, for the last complete step, how the wall time splits into: large kernels (>= 30 us),
small kernels, and time with NO kernel running on the device (launch gaps).  Usage: python tools/timeline_gaps.py kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
# step boundaries: the fused AdamW over the encoder arena is the last large launch of a step (the largest adamw_kernel launch)
ad = [i for i, e in enumerate(ev) if "adamw_kernel" in e[2]]
big_ad = [i for i in ad if ev[i][1] - ev[i][0] > 40000]
assert len(big_ad) >= 3, len(big_ad)
lo, hi = big_ad[-3], big_ad[-2]
seg = ev[lo + 1:hi + 1]
t0, t1 = ev[lo][1], ev[hi][1]
busy = []  # union of kernel intervals
for s, e, _ in seg:
    if busy and s <= busy[-1][1]:
        busy[-1][1] = max(busy[-1][1], e)
    else:
        busy.append([s, e])
covered = sum(e - s for s, e in busy)
small = [(s, e, n) for s, e, n in seg if e - s < 30000]
# time during which ONLY small kernels run: union(small) minus union(large)
large = [(s, e) for s, e, n in seg if e - s >= 30000]
def union(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out
ul = union(large)
cov_large = sum(e - s for s, e in ul)
print(f"step wall {1e-6 * (t1 - t0):.3f} ms: {len(seg)} launches, {len(small)} of them < 30 us")
print(f"  some kernel running          {1e-6 * covered:.3f} ms")
print(f"  a large kernel running       {1e-6 * cov_large:.3f} ms")
print(f"  only small kernels running   {1e-6 * (covered - cov_large):.3f} ms")
print(f"  nothing running (gaps)       {1e-6 * (t1 - t0 - covered):.3f} ms")
gaps = []
for (s0, e0), (s1, e1) in zip(busy, busy[1:]):
    gaps.append((s1 - e0, e0))
gaps.sort(reverse=True)
print("  largest gaps (us):", " ".join(f"{g / 1e3:.1f}" for g, _ in gaps[:12]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:48]
by_end = {e: n for s, e, n in seg}
by_start = {}
for s, e, n in seg:
    by_start.setdefault(s, n)
starts = sorted(by_start)
import bisect
print("  gap   after kernel                                      -> next kernel")
for g, e0 in gaps[:24]:
    nxt = starts[bisect.bisect_right(starts, e0)] if bisect.bisect_right(starts, e0) < len(starts) else None
    print(f"  {g / 1e3:5.1f} {short(by_end.get(e0, '?')):50s} -> {short(by_start.get(nxt, '?'))}")
# small kernels by name: count and time inside the step
import collections
c = collections.Counter(); d = collections.Counter()
for s, e, n in small:
    c[short(n)] += 1; d[short(n)] += e - s
print("  small kernels of the step:")
for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:25]:
    print(f"    {c[k]:4d} x {k:50s} {v / 1e3:8.1f} us")

# phases of the step by landmarks: first / last attention forward launch, first / last attention backward launch
def first(sub):
    return next(s for s, e, n in seg if sub in n)
def last(sub):
    return [e for s, e, n in seg if sub in n][-1]
marks = [("optimiser tail + tokeniser forward (before the first attention forward)", t0, first("attn_fwd")),
         ("encoder forward (first to last attention forward)", first("attn_fwd"), last("attn_fwd")),
         ("end of forward, losses, start of backward (to the first attention backward)", last("attn_fwd"), first("attn_bwd")),
         ("encoder backward (first to last attention backward)", first("attn_bwd"), last("attn_bwd")),
         ("rest of backward, tokeniser backward, optimiser", last("attn_bwd"), t1)]
ul_all = union([(s, e) for s, e, n in seg])
def clip(iv, a, b):
    return sum(max(0, min(e, b) - max(s, a)) for s, e in iv)
print("  phase: wall = large kernels + only-small + gaps   [ms]")
for name, a, b in marks:
    big, anyk = clip(ul, a, b), clip(ul_all, a, b)
    print(f"    {1e-6 * (b - a):7.3f} = {1e-6 * big:7.3f} + {1e-6 * (anyk - big):6.3f} + {1e-6 * (b - a - anyk):6.3f}   {name}")

if len(sys.argv) > 2:  # list the launches of one region in order: "list" = the loss region, "head" = from the step's start to the first attention forward
    a, b = (last("attn_fwd"), first("attn_bwd")) if sys.argv[2] != "head" else (t0, first("attn_fwd"))
    print("  launches of the region (start offset us, duration us, name):")
    for s_, e_, n_ in seg:
        if s_ >= a and s_ < b:
            print(f"    {1e-3 * (s_ - a):8.1f} {1e-3 * (e_ - s_):6.1f}  {short(n_)}")

# gap statistics by the pair (kernel that ended, kernel that started next), for pairs seen at least 5 times
pairs = collections.defaultdict(list)
ends = sorted((e, n) for s_, e, n in seg)
for (s0, e0), (s1, e1) in zip(busy, busy[1:]):
    nxt = starts[bisect.bisect_right(starts, e0)] if bisect.bisect_right(starts, e0) < len(starts) else None
    pairs[(short(by_end.get(e0, "?"))[:28], short(by_start.get(nxt, "?"))[:28])].append(s1 - e0)
print("  idle time between the end of one kernel and the start of the next, by pair (count, median us):")
for k, v in sorted(pairs.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 5:
        v = sorted(v)
        print(f"    {len(v):3d} x {v[len(v) // 2] / 1e3:5.1f} us   {k[0]:28s} -> {k[1]}")
