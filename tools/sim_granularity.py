"""What finer pruning granularity would buy the kNN search: the kernel decides per (wave = 64 queries, tile = 64 references)
whether to compute a 64 x 64 distance block; the MFMA block is 32 x 32.  For sampled waves, with the kernel's own rule (per-query
ball test against the seeds AND the transposed ball test against the wave's largest seed) evaluated per 64q x 64r, 32q x 64r,
64q x 32r and 32q x 32r, print the fraction of the distance work each granularity computes.
python tools/sim_granularity.py [N] [n_waves]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
import meld_amd
from bench import synthetic_cells

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 96
knn, ksel = 15, 64
rf2 = float(np.log(1e4) ** (2.0 / 40.0))
X, _ = synthetic_cells(N, 50, seed=0)
Xd = torch.from_numpy(X).cuda()
op = meld_amd.MELD(knn=knn, verbose=0).fit(Xd)
G = op.graph
Xo = Xd[G.perm]
T = (N + 63) // 64
pad = T * 64 - N
Xp = torch.cat([Xo, Xo[-1:].expand(pad, -1)]) if pad else Xo


def meb(pts, steps=24):
    c0 = pts.mean(1)
    r0 = torch.linalg.vector_norm(pts - c0[:, None, :], dim=2).max(1).values
    cb, rb, cc = c0.clone(), r0.clone(), c0.clone()
    ar = torch.arange(pts.shape[0], device=pts.device)
    for it in range(1, steps + 2):
        dist = torch.linalg.vector_norm(pts - cc[:, None, :], dim=2)
        r, who = dist.max(1)
        better = r < rb
        cb[better], rb[better] = cc[better], r[better]
        if it <= steps:
            cc = cc + (pts[ar, who] - cc) / (it + 1)
    return cb, rb


C1, r1 = meb(Xp.view(T, 64, -1))          # tile spheres
C2, r2 = meb(Xp.view(2 * T, 32, -1))      # half tiles
C4, r4 = meb(Xp.view(4 * T, 16, -1))      # quarter tiles (the ordering's leaves)
print("radius: tile %.3f  half %.3f  quarter %.3f" % (r1.mean(), r2.mean(), r4.mean()))
# a bound that is not a ball (round-4 review, item 1a): the axis-aligned box of a tile in the top-k principal coordinates
# (distance within a subspace <= distance), combined with the ball bound by max
Xc = Xo - Xo.mean(0, keepdim=True)
evals, evecs = torch.linalg.eigh(Xc.T @ Xc / N)
BOXK = (6, 8, 10, 16)
proj = {k: (Xp - Xo.mean(0, keepdim=True)) @ evecs[:, -k:] for k in BOXK}
boxes = {}
for k in BOXK:
    for rs in (1, 2):
        pk = proj[k].view(T * rs, 64 // rs, k)
        boxes[(k, rs)] = (pk.min(1).values, pk.max(1).values)


def box_dist(Pk, lo, hi):
    """[nq, n_boxes] euclidean distance of every query to every box (0 inside)"""
    out = torch.zeros(Pk.shape[0], lo.shape[0], dtype=Pk.dtype, device=Pk.device)
    for j in range(Pk.shape[1]):
        e = torch.clamp_min(torch.maximum(lo[None, :, j] - Pk[:, j:j + 1], Pk[:, j:j + 1] - hi[None, :, j]), 0)
        out += e * e
    return out.sqrt()


g = torch.Generator().manual_seed(0)
waves = torch.randint(32, T - 40, (NW,), generator=g).tolist()
n2 = (Xo * Xo).sum(1)
acc = {}


def add(k, v):
    acc.setdefault(k, []).append(float(v))


for w in waves:
    P = Xo[64 * w: 64 * w + 64]
    d2 = ((P * P).sum(1)[:, None] + n2[None, :] - 2.0 * P @ Xo.T).clamp_min(0)
    srt = torch.topk(d2, ksel, dim=1, largest=False).values
    thr = torch.minimum(srt[:, ksel - 1], rf2 * srt[:, knn])
    lo, hi = max(0, 64 * (w - 32)), min(N, 64 * (w + 36))
    seed = rf2 * torch.topk(d2[:, lo:hi], knn + 1, dim=1, largest=False).values[:, knn]   # the product's +-32 tile window
    s = seed.sqrt()
    hit = d2 < thr[:, None]
    hit_p = torch.cat([hit, hit.new_zeros(64, pad)], 1) if pad else hit
    add("ideal 64q x 64r (block holds a candidate)", hit_p.view(64, T, 64).any(2).any(0).float().mean())
    add("ideal 32q x 32r", hit_p.view(2, 32, T, 2, 32).any(4).any(1).float().mean())
    add("ideal 1q x 64r", hit_p.view(64, T, 64).any(2).float().mean())

    def rule(qsplit, rsplit):
        """live[qg, T * rsplit]: per-query test against the spheres of the reference pieces AND the transposed test (the
        references of a piece against the sphere of the query group, within reach of the group's largest seed)"""
        Cr, rr = {1: (C1, r1), 2: (C2, r2), 4: (C4, r4)}[rsplit]
        Cq, rq = {1: (C1, r1), 2: (C2, r2), 4: (C4, r4)}[qsplit]
        nq = 64 // qsplit
        lb = torch.cdist(P, Cr) - rr[None, :]                       # [64, T * rsplit]
        a = (lb <= s[:, None]).view(qsplit, nq, -1).any(1)          # per query group
        live = []
        for gq in range(qsplit):
            cq, rho = Cq[qsplit * w + gq], rq[qsplit * w + gq]
            dmin = torch.linalg.vector_norm(Xp - cq[None, :], dim=1).view(T * rsplit, 64 // rsplit).min(1).values
            bt = (dmin - rho) <= s[gq * nq:(gq + 1) * nq].max()
            live.append(a[gq] & bt)
        return torch.stack(live)                                    # [qsplit, T * rsplit]

    base = None
    for qs, rs in ((1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (1, 4), (4, 4), (4, 2)):
        lv = rule(qs, rs)
        frac = lv.float().mean()
        add("kernel rule %2dq x %2dr: work computed" % (64 // qs, 64 // rs), frac)
        # tiles some piece of the workgroup's wave needs (what has to be staged for this wave)
        add("   ... tiles touched (any piece live)", lv.view(qs, T, rs).any(2).any(0).float().mean())
    # per-query test alone (no transposed test), per single query
    lb1 = torch.cdist(P, C1) - r1[None, :]
    add("per-query ball test alone, 1q x 64r (mean over queries)", (lb1 <= s[:, None]).float().mean())
    lv11 = rule(1, 1)[0]
    lb2 = torch.cdist(P, C2) - r2[None, :]
    for k in BOXK:
        Pk = proj[k][64 * w: 64 * w + 64]
        bd = box_dist(Pk, *boxes[(k, 1)])
        keep = torch.maximum(lb1, bd) <= s[:, None]
        add("per-query ball + %2d-d principal box, 1q x 64r" % k, keep.float().mean())
        add("   kernel rule 64q x 64r with the box in the per-query test (k = %2d)" % k, (keep.any(0) & lv11).float().mean())
        bd2 = box_dist(Pk, *boxes[(k, 2)])
        keep2 = torch.maximum(lb2, bd2) <= s[:, None]
        add("   per-query ball + box on half tiles, 1q x 32r (k = %2d)" % k, keep2.float().mean())
        add("   64q x 32r pieces with ball + box (k = %2d; no transposed test)" % k, keep2.any(0).float().mean())
wt = G.info.get("wave_tiles_done")
print("N = %d, %d sampled waves; product computes %.4f of the blocks" % (N, NW, (wt / float(T * T)) if wt else float("nan")))
for k, v in acc.items():
    print("  %-62s %.4f" % (k, np.mean(v)))
