"""The wide (lanes = columns) recurrence step against p / 2 launches of the two-column kernel: python tools/time_wide.py [N] [p]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import torch, meld_amd
from meld_amd import filter as mf
from bench import synthetic_cells
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 64
X, _ = synthetic_cells(N, 50, seed=0)
G = meld_amd.MELD(knn=int(os.environ.get("KNN", "15")), verbose=0).fit(torch.from_numpy(X).cuda()).graph  # (the benchmark graph: knn = 15, 39.4 M nonzeros)
ops = mf._ops_of(G)
x = torch.rand(N, p, dtype=torch.float64, device="cuda"); z = torch.rand(N, p, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
xp = x.view(N, p // 2, 2).permute(1, 0, 2).contiguous(); zp = z.view(N, p // 2, 2).permute(1, 0, 2).contiguous(); yp = torch.empty_like(xp)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def wide(): ops.cheby_step_wide(G, p, x, 0, z, y, 0.7, -0.2, -1.0)
def pairs():
    for i in range(p // 2):
        yp[i].copy_(zp[i]); ops.cheby_step(G, 2, xp[i], 0, yp[i], yp[i], None, 0.7, -0.2, -1.0, 0.0)
tw, tp = t(wide), t(pairs)
err = float((y - yp.permute(1, 0, 2).reshape(N, p)).abs().max() / y.abs().max())
byts = 12 * G.nnz + 4 * (N + 1) + 8 * N + 40 * N * p
print("N=%d p=%d: wide %.3f ms (%.2f of HBM by the algorithmic bytes), pairs on the tiled kernel %.3f ms (incl. %d copies); max rel diff %.2e" % (N, p, tw, byts / tw / 1e6 / 8000, tp, p // 2, err))
# as the filter bank drives it: T_k overwrites T_{k-2} in place (y aliases z), buffers alternate
a_, b_ = x.clone(), z.clone()
def pingpong():
    global a_, b_
    ops.cheby_step_wide(G, p, a_, 0, b_, b_, 1e-3, 0.2, -1.0)
    a_, b_ = b_, a_
print("ping-pong, y aliases z: %.3f ms" % t(pingpong, 10))
