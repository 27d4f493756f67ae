"""Python model of csrc/nms.hip nms_fused_kernel's greedy phase (round 5): the reference's pick order with bit-equal scores
(nms.c:74-98: first maximum in ARRAY order; the old first element takes the picked box's slot; survivors keep their order) simulated
exactly with two bitsets — alive-by-RANK (sorted: score desc, index asc) and alive-by-POSITION (slot in the reference's array) —
plus pos[rank] / owner[slot].  Validated here against the reference's compiled nms.c (oracle/_ref) before the kernel was written.
Run:  python tools/models/nms_fused_model.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def iou(a, b):
    f = np.float32
    x1, y1, x2, y2 = max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])
    w, h = f(f(x2 - x1) + f(1)), f(f(y2 - y1) + f(1))
    inter = f(w * h)
    aa = f(f(f(a[2] - a[0]) + f(1)) * f(f(a[3] - a[1]) + f(1)))
    ba = f(f(f(b[2] - b[0]) + f(1)) * f(f(b[3] - b[1]) + f(1)))
    with np.errstate(all="ignore"):
        v = f(inter / f(f(aa + ba) - inter))
    return f(0) if (w <= 0 or h <= 0) else v


def fused(sb, thr):
    sb = np.asarray(sb, np.float32)
    m = sb.shape[0]
    s = sb[:, 4].copy()
    s[s == 0] = 0.0                                     # -0.0 == 0.0 for the reference's '>' : one key
    order = sorted(range(m), key=lambda i: (-float(s[i]), i))
    sid = order
    n_sel = sum(1 for i in range(m) if s[i] > -1e7)
    eq = [r + 1 < m and s[sid[r]] == s[sid[r + 1]] for r in range(m)]
    has_ties = any(eq)
    with np.errstate(all="ignore"):
        sup = [[(c != r) and not (iou(sb[sid[r]], sb[sid[c]]) <= np.float32(thr)) for c in range(m)] for r in range(m)]
    alive_r = [True] * m
    kept = []
    if not has_ties:
        for r in range(n_sel):
            if alive_r[r]:
                kept.append(r)
                for c in range(r + 1, m):
                    if sup[r][c]:
                        alive_r[c] = False
        return sb[[sid[r] for r in kept]], [sid[r] for r in kept]
    pos = list(sid)                                     # pos[rank] = slot; initially the original index
    owner = [0] * m
    for r in range(m):
        owner[sid[r]] = r
    alive_p = [True] * m
    while True:
        r0 = next((r for r in range(m) if alive_r[r]), None)
        if r0 is None or r0 >= n_sel:
            break
        b = r0
        if eq[r0]:
            e = r0
            while eq[e]:
                e += 1
            b = owner[min(pos[r] for r in range(r0, e + 1) if alive_r[r])]
        pf = next(p for p in range(m) if alive_p[p])    # the head: boxes[0]
        pb = pos[b]
        if pf != pb:                                    # nms.c:83-85 swap: the head takes the pick's slot
            f = owner[pf]
            owner[pb] = f
            pos[f] = pb
            alive_p[pf] = False
        else:
            alive_p[pb] = False
        kept.append(b)
        alive_r[b] = False
        for c in range(m):
            if alive_r[c] and sup[b][c]:
                alive_r[c] = False
                alive_p[pos[c]] = False
    return sb[[sid[r] for r in kept]], [sid[r] for r in kept]


def fused_v2(sb, thr):
    """the kernel's SECOND form (what ships): occupied-by-position bits dropped LAZILY (a dead owner is skipped when it surfaces as the head),
    the current run's slots cached (`mp`) and patched when the head that moves is a member — same picks as fused()"""
    sb = np.asarray(sb, np.float32)
    m = sb.shape[0]
    s = sb[:, 4].copy()
    s[s == 0] = 0.0
    s[np.isnan(s)] = -np.inf
    order = sorted(range(m), key=lambda i: (-float(s[i]), i))
    sid = order
    n_sel = sum(1 for i in range(m) if s[i] > -1e7)
    eq = [r + 1 < m and s[sid[r]] == s[sid[r + 1]] for r in range(m)]
    if not any(eq):
        return fused(sb, thr)
    with np.errstate(all="ignore"):
        sup = [[(c != r) and not (iou(sb[sid[r]], sb[sid[c]]) <= np.float32(thr)) for c in range(m)] for r in range(m)]
    alive = [True] * m
    pos = list(sid)
    owner = [0] * m
    for r in range(m):
        owner[sid[r]] = r
    occ = [True] * m
    run_s, run_e, mp = 0, -1, {}
    kept = []
    while True:
        r0 = next((r for r in range(m) if alive[r]), None)
        if r0 is None or r0 >= n_sel:
            break
        b, pb = r0, None
        if eq[r0]:
            e = r0
            while eq[e]:
                e += 1
            if e - r0 < 64:
                if e != run_e:
                    run_s, run_e = r0, e
                    mp = {r: pos[r] for r in range(run_s, e + 1)}
                best = min(mp[r] for r in range(run_s, run_e + 1) if alive[r])
                b = next(r for r in range(run_s, run_e + 1) if alive[r] and mp[r] == best)
                pb = best
            else:
                best = min(pos[r] for r in range(r0, e + 1) if alive[r])
                b, pb = owner[best], best
        if pb is None:
            pb = pos[b]
        while True:
            pf = next(p for p in range(m) if occ[p])
            f = owner[pf]
            if alive[f]:
                break
            occ[pf] = False
        occ[pf] = False
        if pf != pb:
            owner[pb] = f
            pos[f] = pb
            occ[pb] = True
            if run_s <= f <= run_e:
                mp[f] = pb
        kept.append(b)
        alive[b] = False
        for c in range(m):
            if alive[c] and sup[b][c]:
                alive[c] = False
    return sb[[sid[r] for r in kept]], [sid[r] for r in kept]


def main():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import random_scored_boxes
    from oracle import mpn_oracle as O
    O.build()
    n_cases = 0
    for regime in ("distinct", "ties", "saturated", "allequal"):
        for m in (1, 2, 3, 17, 64, 65, 130, 200, 300):
            for seed in range(4):
                rng = np.random.default_rng(seed * 1000 + m)
                sb = random_scored_boxes(rng, m, regime, span=300.0, lo=16.0, hi=200.0)
                unpick = seed == 3 and m > 3
                if seed == 2 and m > 3:                 # a signed zero among zeros: equal for the reference's '>'
                    sb[rng.integers(0, m, 2), 4] = 0.0
                    sb[rng.integers(0, m, 1), 4] = -0.0
                if unpick:                              # rows nms.c:75 never picks (with them the compiled reference runs into best = -1: UB;
                    sb[rng.integers(0, m, 3), 4] = -2e7  # the oracle's restatement defines the behaviour: stop)
                ref = O.nms(sb, 0.3) if unpick else O.ref_nms(sb, 0.3)
                for fn in (fused, fused_v2):
                    got, _ = fn(sb, 0.3)
                    assert np.array_equal(got, ref), (fn.__name__, regime, m, seed)
                n_cases += 1
    print("nms_fused_model: %d cases == compiled nms.c" % n_cases)


if __name__ == "__main__":
    main()
