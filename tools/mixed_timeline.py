"""post-process a rocprofv3 --kernel-trace --memory-copy-trace run of `bench.py --mixed-sizes`: the LAST `n` seconds of the timeline —
GPU busy fraction, copy time and how much of it overlaps kernels, the largest idle gaps and what follows them.  python tools/mixed_timeline.py <dir> [seconds]"""
import csv, glob, os, sys
d = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
mc = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "")) for r in csv.DictReader(open(kt))]
C = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", ""), int(r.get("Bytes", 0) or 0)) for r in csv.DictReader(open(mc[0]))] if mc else []
K.sort()
t1 = K[-1][1]; t0 = t1 - int(win * 1e9)
Kw = [k for k in K if k[0] >= t0]
Cw = [c for c in C if c[0] >= t0]
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    out = []
    for s, e, *_ in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: out.append((cs, ce)); cs, ce = s, e
    if cs is not None: out.append((cs, ce))
    return out
ku = union(Kw)
busy = sum(e - s for s, e in ku)
print("window %.3f s: %d kernels, GPU busy %.1f %%; queues used: %s" % (win, len(Kw), 100.0 * busy / (t1 - t0), sorted(set(k[3] for k in Kw))))
ctot = sum(e - s for s, e, *_ in Cw)
def overlap(a, b):
    i = j = 0; tot = 0
    while i < len(a) and j < len(b):
        s = max(a[i][0], b[j][0]); e = min(a[i][1], b[j][1])
        if e > s: tot += e - s
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return tot
cu = union(Cw)
print("copies: %d, %.1f MB, busy %.1f %% of the window, of which %.1f %% under kernels; mean rate %.1f GB/s" % (
    len(Cw), sum(c[3] for c in Cw) / 1e6, 100.0 * ctot / (t1 - t0), 100.0 * overlap(cu, ku) / max(1, sum(e - s for s, e in cu)), sum(c[3] for c in Cw) / max(1, ctot)))
gaps = sorted(((ku[i + 1][0] - ku[i][1], ku[i][1], ku[i + 1][0]) for i in range(len(ku) - 1)), reverse=True)[:12]
for g, a, b in gaps:
    nxt = next(k for k in Kw if k[0] == b)
    prv = max((k for k in Kw if k[1] <= a), key=lambda k: k[1])
    cin = [c for c in Cw if c[0] < b and c[1] > a]
    print("  idle %7.1f us after %-40s before %-40s copies in the gap: %s" % (g / 1e3, prv[2], nxt[2], ["%.1fMB %.0fus" % (c[3] / 1e6, (c[1] - c[0]) / 1e3) for c in cin]))
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q in Kw:
    a = agg[(n, q)]; a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print("sum of kernel durations = %.2f x the window (>1: kernels of different queues overlap)" % (tot * 1e3 / (t1 - t0)))
for (n, q), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-42s queue %-3s calls %5d avg %8.1f us total %6.1f %%" % (n, q, c, us / c, 100.0 * us / tot))
byq = collections.defaultdict(lambda: collections.Counter())
for s, e, n, q in Kw:
    byq[q][n] += 1
for q in sorted(byq):
    print("queue %s: %s" % (q, ", ".join("%s x%d" % (n.split("::")[-1][:28], c) for n, c in byq[q].most_common(8))))
