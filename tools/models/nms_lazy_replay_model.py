"""Python model of the chunked NMS scan with LAZY, VECTORISED position replay (nms.hip: nms_scan_kernel, flag 3), validated here
against the reference's own nms.c (oracle/_ref) before the kernel was written.  CPU only; not part of the product.

The reference's winner among bit-equal scores is the member of the run that sits FIRST in its array (nms.c:74-81), and the array is
permuted every round: the old head takes the pick's place (nms.c:83-85), survivors keep their order (nms.c:91-98).  The kernel
  * resolves 64-rank chunks tie-free (pick order = rank order) as long as no equal-score run has two alive members at its turn;
  * when a run with >= 2 alive members comes up, it first brings the slot model up to date — `simulate()` below: the heads of all
    rounds since the last update, 64 slots per window, the heads of consecutive rounds found as a 64-lane fixpoint
        taken_l = valid_l and death_l >= t0 + popcount(taken & lanes_below_l)
    with the batch cut short where a move lands inside the window ahead of the head or a pick was itself moved earlier in the batch;
  * death rounds (alive-at-round tests) are computed in one data-parallel pass from the symmetric mask rows and the kept list."""
import numpy as np

FOREVER = 1 << 30


def iou(a, b):
    x1, y1, x2, y2 = max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])
    w, h = np.float32(x2 - x1 + np.float32(1)), np.float32(y2 - y1 + np.float32(1))
    inter = np.float32(w * h)
    aa = np.float32(np.float32(a[2] - a[0] + np.float32(1)) * np.float32(a[3] - a[1] + np.float32(1)))
    ab = np.float32(np.float32(b[2] - b[0] + np.float32(1)) * np.float32(b[3] - b[1] + np.float32(1)))
    if w <= 0 or h <= 0:
        return np.float32(0)
    return np.float32(inter / np.float32(np.float32(aa + ab) - inter))


def model_nms(sb, thr, stats=None, V2=False):
    sb = np.asarray(sb, np.float32)
    m = sb.shape[0]
    order = sorted(range(m), key=lambda i: (-sb[i, 4], i))          # (score desc, index asc) = nms_sort_kernel
    rank_box = sb[order]
    sc = rank_box[:, 4]
    n = int((sc > -1e7).sum())
    # symmetric mask rows (flag 3: full), nms.c:14-41 in float32, vectorised
    f = np.float32
    x1 = np.maximum(rank_box[:, None, 0], rank_box[None, :, 0]); y1 = np.maximum(rank_box[:, None, 1], rank_box[None, :, 1])
    x2 = np.minimum(rank_box[:, None, 2], rank_box[None, :, 2]); y2 = np.minimum(rank_box[:, None, 3], rank_box[None, :, 3])
    w = (x2 - x1 + f(1)).astype(f); h = (y2 - y1 + f(1)).astype(f)
    inter = (w * h).astype(f)
    area = ((rank_box[:, 2] - rank_box[:, 0] + f(1)).astype(f) * (rank_box[:, 3] - rank_box[:, 1] + f(1)).astype(f)).astype(f)
    with np.errstate(all="ignore"):
        io = (inter / ((area[:, None] + area[None, :]).astype(f) - inter).astype(f)).astype(f)
    io = np.where((w <= 0) | (h <= 0), f(0), io)
    over = ~(io <= f(thr))
    np.fill_diagonal(over, False)
    tie = np.zeros(m, bool)                                          # bit r: ranks r and r+1 pickable with equal scores
    for r in range(n - 1):
        tie[r] = sc[r] == sc[r + 1]
    removed = np.zeros(m, bool)
    pos = np.array(order, np.int64).copy()                           # pos[rank] = slot (original index)
    occ = np.full(m, -1, np.int64)
    for r in range(m):
        occ[pos[r]] = r
    kept = []                                                        # kept[t-1] = rank picked in round t
    round_of = np.full(m, -1, np.int64)
    sim = dict(done=0, hp=0)                                         # rounds already replayed, head pointer

    def deaths():
        d = np.full(m, FOREVER, np.int64)
        for j in range(m):
            if round_of[j] > 0:
                d[j] = round_of[j]
            elif removed[j]:
                ks = [round_of[k] for k in kept if over[j, k]]
                d[j] = min(ks)
        return d

    def simulate():
        """replay rounds sim.done+1 .. len(kept) on the slot model, window by window"""
        t1 = len(kept)
        if sim["done"] >= t1:
            return
        d = deaths()
        if stats is not None:
            stats["sims"] = stats.get("sims", 0) + 1
        while sim["done"] < t1:
            W0 = sim["hp"] & ~63
            p = sim["hp"] - W0
            f = [int(occ[W0 + l]) if W0 + l < m else -1 for l in range(64)]
            dl = [int(d[f[l]]) if f[l] >= 0 else -1 for l in range(64)]
            t0 = sim["done"] + 1
            # fixpoint over the window
            taken = [False] * 64
            for _ in range(66):
                new = []
                cnt = 0
                below = 0
                for l in range(64):
                    rd = t0 + sum(1 for q in range(l) if taken[q])
                    new.append(l >= p and f[l] >= 0 and dl[l] >= rd and rd <= t1)
                if new == taken:
                    break
                taken = new
            else:
                raise AssertionError("fixpoint did not settle")
            if stats is not None:
                stats["batches"] = stats.get("batches", 0) + 1
            lanes = [l for l in range(64) if taken[l]]
            if not lanes:
                sim["hp"] = W0 + 64
                assert sim["hp"] < m + 64, "ran out of slots with rounds pending"
                continue
            # commit in round order, cutting the batch at a hazard
            moved = set()
            last = None
            for k, l in enumerate(lanes):
                t = t0 + k
                pick = kept[t - 1]
                if pick in moved:          # H1: the pick was relocated earlier in this batch: its slot is stale in LDS
                    break
                last = l
                sim["done"] = t
                if f[l] != pick:
                    sbslot = int(pos[pick])
                    assert sbslot >= W0 + l
                    occ[sbslot] = f[l]
                    pos[f[l]] = sbslot
                    moved.add(f[l])
                    if sbslot < W0 + 64:   # H2: the move lands inside the window ahead of the head: later rounds see a new occupant
                        break
            sim["hp"] = W0 + (last + 1 if last is not None else p)
            if last is None:               # first taken lane already hazardous (cannot happen: moved is empty) -> defensive
                raise AssertionError("empty commit")

    def simulate_v2():
        """round 5 (nms_fused_kernel's replay): a move that lands INSIDE the window no longer cuts the batch.  The 64 rounds a window can
        hold are preloaded in 'round space' (lane j: the pick of round t0 + j, its slot, the window lane that slot is — the landing lane of
        round j's head); the fixpoint then iterates over (taken, occupant, death) together: the head of round j is the j-th taken lane, lane
        X's occupant is the head of the round landing on it if that head sits below X.  Only a pick that was itself moved earlier in the
        batch (stale slot) still cuts."""
        t1 = len(kept)
        if sim["done"] >= t1:
            return
        d = deaths()
        if stats is not None:
            stats["sims"] = stats.get("sims", 0) + 1
        while sim["done"] < t1:
            W0 = sim["hp"] & ~63
            p = sim["hp"] - W0
            f0 = [int(occ[W0 + l]) if W0 + l < m else -1 for l in range(64)]
            d0 = [int(d[f0[l]]) if f0[l] >= 0 else -1 for l in range(64)]
            t0 = sim["done"] + 1
            pk = [kept[t0 + j - 1] if t0 + j <= t1 else -1 for j in range(64)]
            sbj = [int(pos[pk[j]]) if pk[j] >= 0 else -1 for j in range(64)]
            jr = [-1] * 64
            for j in range(64):
                if pk[j] >= 0 and W0 <= sbj[j] < W0 + 64:
                    assert jr[sbj[j] - W0] < 0
                    jr[sbj[j] - W0] = j
            taken = [False] * 64
            f, dd = list(f0), list(d0)
            for it in range(70):
                k = [sum(1 for q in range(l) if taken[q]) for l in range(64)]
                cnt = sum(taken)
                R = [None] * 64
                for l in range(64):
                    if taken[l]:
                        R[k[l]] = (l, f[l], dd[l])
                nf, nd = list(f0), list(d0)
                for x in range(64):
                    j = jr[x]
                    if j >= 0 and j < cnt and R[j][0] < x:
                        nf[x], nd[x] = R[j][1], R[j][2]
                nt = [l >= p and nf[l] >= 0 and nd[l] >= t0 + k[l] and t0 + k[l] <= t1 for l in range(64)]
                if nt == taken and nf == f and nd == dd:
                    break
                taken, f, dd = nt, nf, nd
            else:
                raise AssertionError("fixpoint did not settle")
            if stats is not None:
                stats["batches"] = stats.get("batches", 0) + 1
                stats["iters"] = stats.get("iters", 0) + it + 1
            k = [sum(1 for q in range(l) if taken[q]) for l in range(64)]
            cnt = sum(taken)
            if cnt == 0:
                sim["hp"] = W0 + 64
                assert sim["hp"] < m + 64, "ran out of slots with rounds pending"
                continue
            R = [None] * 64
            for l in range(64):
                if taken[l]:
                    R[k[l]] = (l, f[l], dd[l])
            moved = set(R[j][1] for j in range(cnt) if R[j][1] != pk[j])
            cut = cnt
            for j in range(cnt):
                if pk[j] in moved:      # stale slot: the pick was a head (and moved) earlier in this batch
                    cut = j
                    if stats is not None:
                        stats["h1"] = stats.get("h1", 0) + 1
                    break
            assert cut > 0
            for j in range(cut):
                hl, hf, hd = R[j]
                if hf != pk[j]:
                    assert sbj[j] > W0 + hl
                    occ[sbj[j]] = hf
                    pos[hf] = sbj[j]
            sim["done"] += cut
            if cut == cnt and sim["done"] < t1:
                sim["hp"] = W0 + 64       # every lane above the last head is dead on arrival
            else:
                sim["hp"] = W0 + R[cut - 1][0] + 1

    if V2:
        simulate = simulate_v2

    def pick_rank(r, by_rule=False):
        t = len(kept) + 1
        kept.append(r)
        round_of[r] = t
        removed[r] = True
        removed[over[r]] = True

    nchunks = (n + 63) // 64
    for c in range(nchunks):
        base = c * 64
        nv = min(64, n - base)
        while True:
            alive = [base + l for l in range(nv) if not removed[base + l]]
            if not alive:
                break
            # first alive rank whose equal-score run has ANOTHER alive member
            limit = None
            for r in alive:
                # run of r: extend both ways over tie bits
                lo = r
                while lo > 0 and tie[lo - 1]:
                    lo -= 1
                hi = r
                while hi < n - 1 and tie[hi]:
                    hi += 1
                members = [q for q in range(lo, hi + 1) if not removed[q]]
                if len(members) >= 2:
                    limit = r
                    run = members
                    break
            # tie-free rule for alive ranks below `limit`: greedy in rank order (the kernel's 64-lane fixpoint)
            if limit is None or alive[0] < limit:
                for r in alive:
                    if limit is not None and r >= limit:
                        break
                    if not removed[r]:
                        pick_rank(r)
                if limit is None:
                    break
                continue   # the picks above may have removed members of the run: look again
            # exact rule for one pick: the alive member of the run sitting first in the array.  While a run is being picked its members'
            # slots cannot change — the head that moves in a round is either a non-member (a lower score sitting earlier in the array) or
            # the pick itself — so the slot model is brought up to date once per RUN (V2), not once per pick.
            run_id = max(run)
            while run_id < n - 1 and tie[run_id]:
                run_id += 1
            if not (V2 and sim.get("run") == run_id):
                simulate()
                sim["run"] = run_id
            best = min(run, key=lambda q: pos[q])
            if stats is not None:
                stats["exact"] = stats.get("exact", 0) + 1
            pick_rank(best)
    keep = rank_box[kept] if kept else np.zeros((0, 5), np.float32)
    idx = np.array([order[r] for r in kept], np.int64)
    return keep, idx


if __name__ == "__main__":
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import random_scored_boxes
    from oracle import mpn_oracle as O
    O.build()
    rng = np.random.default_rng(0)
    tot = 0
    for regime in ("distinct", "ties", "saturated", "allequal", "fewties"):
        for n in (1, 2, 5, 63, 64, 65, 130, 300):
            for rep in range(3):
                sb = random_scored_boxes(rng, n, "distinct" if regime == "fewties" else regime, span=300.0)
                if regime == "fewties" and n >= 5:
                    for _ in range(3):  # a few bit-equal pairs, one of them a duplicated box
                        a, b = rng.choice(n, 2, replace=False)
                        sb[b, 4] = sb[a, 4]
                    a, b = rng.choice(n, 2, replace=False)
                    sb[b] = sb[a]
                for thr in (0.3, 0.5):
                    ref, ridx = O.nms(sb, thr, return_index=True)
                    if O.have_ref():
                        assert np.array_equal(O.ref_nms(sb, thr), ref)
                    st = {}
                    for v2 in (False, True):
                        k, i = model_nms(sb, thr, st, V2=v2)
                        assert np.array_equal(k, ref) and np.array_equal(i, ridx), (regime, n, rep, thr, v2)
                    tot += 1
        print(regime, "ok")
    print("model == nms.c on", tot, "cases")
