"""How tight could the tile pruning of the kNN search be?  For sampled waves (64 consecutive cells in the
locality order) count the reference tiles a rule keeps, against the tiles that really hold a candidate.
python tools/sim_prune_rules.py [N] [n_waves]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
import meld_amd
from bench import synthetic_cells

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 128
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
tiles = Xp.view(T, 64, -1)
C = tiles.mean(1)
rho_centroid = torch.linalg.vector_norm(tiles - C[:, None, :], dim=2).max(1).values
# centres near the smallest enclosing ball (what tile_spheres_kernel does): 24 Badoiu-Clarkson steps from the centroid
Cb, rb = C.clone(), rho_centroid.clone()
Cc = C.clone()
for it in range(1, 26):
    dist = torch.linalg.vector_norm(tiles - Cc[:, None, :], dim=2)
    r, who = dist.max(1)
    better = r < rb
    Cb[better], rb[better] = Cc[better], r[better]
    if it <= 24:
        far = tiles[torch.arange(T, device=tiles.device), who]
        Cc = Cc + (far - Cc) / (it + 1)
print("tile radius: centroid %.4f -> enclosing-ball centre %.4f (mean)" % (rho_centroid.mean(), rb.mean()))
C, rho = Cb, rb
# the same for half tiles (32 cells)
def meb(pts):
    T2 = pts.shape[0]
    c0 = pts.mean(1); r0 = torch.linalg.vector_norm(pts - c0[:, None, :], dim=2).max(1).values
    cb_, rb_, cc = c0.clone(), r0.clone(), c0.clone()
    for it in range(1, 26):
        dist = torch.linalg.vector_norm(pts - cc[:, None, :], dim=2)
        r, who = dist.max(1)
        better = r < rb_
        cb_[better], rb_[better] = cc[better], r[better]
        if it <= 24:
            far = pts[torch.arange(T2, device=pts.device), who]
            cc = cc + (far - cc) / (it + 1)
    return cb_, rb_
Ch, rh = meb(tiles.reshape(2 * T, 32, -1))
print("half-tile radius (enclosing-ball centres): %.4f (mean)" % rh.mean())
# two half-tile spheres (32 + 32 cells in index order)
C2 = tiles.view(T, 2, 32, -1).mean(2)
rho2 = torch.linalg.vector_norm(tiles.view(T, 2, 32, -1) - C2[:, :, None, :], dim=3).max(2).values
g = torch.Generator().manual_seed(0)
waves = torch.randint(0, T - 1, (NW,), generator=g).tolist()
acc = {}
def add(k, v):
    acc.setdefault(k, []).append(float(v))
n2 = (Xo * Xo).sum(1)
for w in waves:
    P = Xo[64 * w: 64 * w + 64]
    d2 = ((P * P).sum(1)[:, None] + n2[None, :] - 2.0 * P @ Xo.T).clamp_min(0)
    srt = torch.topk(d2, ksel, dim=1, largest=False).values
    thr = torch.minimum(srt[:, ksel - 1], rf2 * srt[:, knn])          # final thresholds (error allowance ignored)
    seed3 = rf2 * torch.topk(d2[:, max(0, 64 * (w - 32)):min(N, 64 * (w + 36))], knn + 1, dim=1, largest=False).values[:, knn]
    lo, hi = max(0, 64 * (w - 4)), min(N, 64 * (w + 8))
    seed = rf2 * torch.topk(d2[:, lo:hi], knn + 1, dim=1, largest=False).values[:, knn]
    # which tiles hold a candidate
    hit = (d2 < thr[:, None])
    hit_p = torch.cat([hit, hit.new_zeros(64, pad)], 1) if pad else hit
    need = hit_p.view(64, T, 64).any(2).any(0)
    Dc = torch.cdist(P, C)                     # 64 x T
    lb = (Dc - rho[None, :]).clamp_min(0)
    lbmin2 = lb.min(0).values ** 2
    add("ideal (tile holds a candidate)", need.float().mean())
    # the transposed bound: every reference of tile t against THIS wave's sphere
    dcw = torch.linalg.vector_norm(Xp - C[w][None, :], dim=1).view(T, 64).min(1).values
    lbT = (dcw - rho[w]).clamp_min(0)
    lbsym2 = torch.maximum(lb.min(0).values, lbT) ** 2
    add("A: min_p LB <= max_p thr (final thresholds)", (lbmin2 <= thr.max()).float().mean())
    add("A: ... with the seeds", (lbmin2 <= seed.max()).float().mean())
    add("B: any_p LB_p <= thr_p (final)", (lb <= thr.sqrt()[:, None]).any(0).float().mean())
    add("B: ... with the seeds", (lb <= seed.sqrt()[:, None]).any(0).float().mean())
    Dc2 = torch.cdist(P, C2.reshape(2 * T, -1)).view(64, T, 2)
    lbh = (Dc2 - rho2[None]).clamp_min(0)
    add("C: half-tile spheres, max thr", ((lbh.min(0).values.min(1).values) ** 2 <= thr.max()).float().mean())
    add("D: half-tile spheres, per query", (lbh.min(2).values <= thr.sqrt()[:, None]).any(0).float().mean())
    phis = ((lb / seed.sqrt()[:, None]).min(0).values)            # smallest relative radius at which the tile is live
    phi = (thr.sqrt() / seed.sqrt()).max()
    add("E: phi rule  min_p LB_p/s_p <= max_p r_p/s_p (final)", (phis <= phi).float().mean())
    add("F: A(final) and B(seeds)", ((lbmin2 <= thr.max()) & (lb <= seed.sqrt()[:, None]).any(0)).float().mean())
    lbH = (torch.cdist(P, Ch).view(64, T, 2) - rh.view(1, T, 2)).clamp_min(0).min(2).values   # nearer of the two half spheres
    add("two half-tile spheres: A(final) and B(seeds +-32)", ((lbH.min(0).values ** 2 <= thr.max()) & (lbH <= seed3.sqrt()[:, None]).any(0)).float().mean())
    add("one sphere:            A(final) and B(seeds +-32)", ((lbmin2 <= thr.max()) & (lb <= seed3.sqrt()[:, None]).any(0)).float().mean())
    add("F + transposed bound in A (final)", ((lbsym2 <= thr.max()) & (lb <= seed.sqrt()[:, None]).any(0)).float().mean())
    add("F + transposed bound in A (seeds)", ((lbsym2 <= seed.max()) & (lb <= seed.sqrt()[:, None]).any(0)).float().mean())
    # transposed per-query test: dead if min_r |r - c_w| > max_p (s_p + |p - c_w|)
    dpc = torch.linalg.vector_norm(P - C[w][None, :], dim=1)
    for nm, sd in (("seeds +-4", seed), ("seeds +-32", seed3)):
        kap = (sd.sqrt() + dpc).max()
        liveT = dcw <= kap
        add("F(sym A final) + transposed per-query test, %s" % nm, ((lbsym2 <= thr.max()) & (lb <= sd.sqrt()[:, None]).any(0) & liveT).float().mean())
        add("   same at the start (A with the seeds), %s" % nm, ((lbsym2 <= sd.max()) & (lb <= sd.sqrt()[:, None]).any(0) & liveT).float().mean())
    add("G: A(final) and E", ((lbmin2 <= thr.max()) & (phis <= phi)).float().mean())
    add("G at the start: A(seeds) and B(seeds)", ((lbmin2 <= seed.max()) & (lb <= seed.sqrt()[:, None]).any(0)).float().mean())
    lo2, hi2 = max(0, 64 * (w - 16)), min(N, 64 * (w + 20))
    seed2 = rf2 * torch.topk(d2[:, lo2:hi2], knn + 1, dim=1, largest=False).values[:, knn]
    add("B with seeds from +-16 tiles", (lb <= seed2.sqrt()[:, None]).any(0).float().mean())
    add("phi (final max r/s)", phi)
    for K in (12, 24, 48, 96):
        near = torch.topk(torch.cdist(C[w:w + 1], C)[0], K, largest=False).indices      # nearest tiles by centroid distance
        cols = (near[:, None] * 64 + torch.arange(64, device=near.device)[None, :]).reshape(-1)
        cols = cols[cols < N]
        seedK = rf2 * torch.topk(d2[:, cols], knn + 1, dim=1, largest=False).values[:, knn]
        add("F with seeds from the %d nearest tiles (centroid distance)" % K, ((lbmin2 <= thr.max()) & (lb <= seedK.sqrt()[:, None]).any(0)).float().mean())
        add("   seed/thr median ratio, %d nearest" % K, (seedK / thr).median())
    add("   seed/thr median ratio, +-4 index tiles", (seed / thr).median())
    for side in (16, 32, 64):
        lo3, hi3 = max(0, 64 * (w - side)), min(N, 64 * (w + side + 4))
        seed3 = rf2 * torch.topk(d2[:, lo3:hi3], knn + 1, dim=1, largest=False).values[:, knn]
        add("F with seeds from +-%d index tiles" % side, ((lbmin2 <= thr.max()) & (lb <= seed3.sqrt()[:, None]).any(0)).float().mean())
        add("   seed/thr median ratio, +-%d index tiles" % side, (seed3 / thr).median())
    # finer granularity: the MFMA block is 32 queries x 32 references -- per (query group, half tile) liveness
    liveB = (lb <= seed3.sqrt()[:, None])                                  # [64 queries, T] full tiles, per query
    add("F(+-32 seeds) 64q x 64r (the kernel's granularity)", ((lbmin2 <= thr.max()) & liveB.any(0)).float().mean())
    lvg = torch.stack([liveB[:32].any(0), liveB[32:].any(0)])              # [2 groups, T]
    add("   32q x 64r: blocks live", lvg.float().mean())
    liveBh = (lbh <= seed3.sqrt()[:, None, None])                          # [64, T, 2] half tiles
    add("   64q x 32r: blocks live", liveBh.any(0).float().mean())
    lvgh = torch.stack([liveBh[:32].any(0), liveBh[32:].any(0)])
    add("   32q x 32r: blocks live", lvgh.float().mean())
    for q in (0.9, 0.75):
        tq = torch.quantile(thr, q)
        add("A with the %.2f quantile of thr instead of the max (not exact)" % q, (lbmin2 <= tq).float().mean())
    add("thr max / median", thr.max() / thr.median())
    add("seed max / thr max", seed.max() / thr.max())
    add("rho / sqrt(median thr)", rho[w] / thr.median().sqrt())
print("N = %d, %d sampled waves; product computes %.3f of the blocks" % (N, NW, G.info.get("blocks_computed_frac", float("nan"))))
for k, v in acc.items():
    print("  %-70s %.4f" % (k, np.mean(v)))
