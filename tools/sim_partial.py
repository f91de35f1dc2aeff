"""Would a partial-distance test pay in the kNN search?  The search computes |r|^2 - 2 q.r for a (wave, tile) block in KB = 4 K blocks of
16 slots.  A distance over a SUBSET of orthonormal coordinates never exceeds the distance, so after the first K block a block whose
partial distances all exceed the rows' thresholds can drop its other three K blocks -- if the first K block holds the coordinates that
carry the distance.  For sampled waves: of the (wave, 32-reference half tile) segments the kernel's own rule computes, the fraction that
would go on after a first K block of the top-k principal coordinates (thresholds = the seeds, i.e. the start values; and the final ones).
python tools/sim_partial.py [N] [n_waves]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
import meld_amd
from bench import synthetic_cells

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 64
DATA = os.environ.get("DATA", "mixture")
knn, ksel = 15, 64
rf2 = float(np.log(1e4) ** (2.0 / 40.0))
if DATA == "iid":
    X = np.random.default_rng(0).normal(size=(N, 50))
else:
    X, _ = synthetic_cells(N, 50, seed=0)
Xd = torch.from_numpy(X).cuda()
op = meld_amd.MELD(knn=knn, verbose=0).fit(Xd)
G = op.graph
Xo = Xd[G.perm]
T = (N + 63) // 64
pad = T * 64 - N
Xp = torch.cat([Xo, Xo[-1:].expand(pad, -1)]) if pad else Xo
tiles = Xp.view(T, 64, -1)
C1 = tiles.mean(1)
r1 = torch.linalg.vector_norm(tiles - C1[:, None, :], dim=2).max(1).values
mean = Xo.mean(0, keepdim=True)
Xc = Xo - mean
evals, evecs = torch.linalg.eigh(Xc.T @ Xc / N)
print("N = %d, data %s; variance in the top 13 / 16 principal coordinates: %.4f / %.4f" % (N, DATA, float(evals[-13:].sum() / evals.sum()), float(evals[-16:].sum() / evals.sum())))
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
    thr_final = torch.minimum(srt[:, ksel - 1], rf2 * srt[:, knn])
    lo, hi = max(0, 64 * (w - 32)), min(N, 64 * (w + 36))
    thr_seed = rf2 * torch.topk(d2[:, lo:hi], knn + 1, dim=1, largest=False).values[:, knn]
    s = thr_seed.sqrt()
    # the kernel's rule per (wave, tile)
    lb = torch.cdist(P, C1) - r1[None, :]
    a = (lb <= s[:, None]).any(0)
    dmin = torch.linalg.vector_norm(Xp - C1[w][None, :], dim=1).view(T, 64).min(1).values
    live = a & ((dmin - r1[w]) <= s.max())                        # [T] tiles this wave computes
    add("computed (wave, tile) blocks", live.float().mean())
    d2p = torch.cat([d2, d2.new_full((64, pad), float("inf"))], 1) if pad else d2
    for name, thr in (("seed", thr_seed), ("final", thr_final)):
        hit = (d2p < thr[:, None]).view(64, T, 2, 32)
        add("  blocks with a candidate under the %s thresholds: 64r / 32r pieces (of all)" % name, hit.any(3).any(2).any(0)[live].float().sum() / T)
    for k in (8, 10, 13, 16, 29):
        V = evecs[:, -k:]
        Pa = (P - mean) @ V
        Xa = (Xp - mean) @ V
        d2a = ((Pa * Pa).sum(1)[:, None] + (Xa * Xa).sum(1)[None, :] - 2.0 * Pa @ Xa.T).clamp_min(0)
        for name, thr in (("seed", thr_seed), ("final", thr_final)):
            go = (d2a < thr[:, None]).view(64, T, 2, 32).any(3).any(0)      # [T, 2]: the half tile goes on after the first K block
            frac32 = go[live].float().mean()                                  # of the computed 32-ref segments
            frac64 = go.any(1)[live].float().mean()
            add("  first K block = top %2d principal coords, %s thresholds: segments (64q x 32r) that go on / tiles with one" % (k, name), frac32)
            add("     ... 64r", frac64)
for k, v in acc.items():
    print("  %-110s %.4f" % (k, np.mean(v)))
