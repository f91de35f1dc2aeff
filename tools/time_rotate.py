"""Where the principal-frame rotation of the search stage spends its time (1M x 50)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synthetic_cells
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X = torch.from_numpy(synthetic_cells(n, 50, seed=0)[0]).cuda()
mean = X.mean(0)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out
for step in (15, 61):
    ms, Xc = t(lambda: X[::step] - mean); print("sample step %d: %.3f ms" % (step, ms))
    ms, cov = t(lambda: Xc.T @ Xc); print("  cov mm: %.3f ms" % ms)
ms, covh = t(lambda: cov.cpu().numpy()); print("cov D2H: %.3f ms" % ms)
ms, (ev, evec) = t(lambda: np.linalg.eigh(covh)); print("numpy eigh 50x50: %.3f ms" % ms)
ms, _ = t(lambda: torch.linalg.eigh(cov)); print("torch eigh on device: %.3f ms" % ms)
ms, V = t(lambda: torch.from_numpy(np.ascontiguousarray(evec[:, ::-1])).to(X.device)); print("V H2D: %.3f ms" % ms)
ms, Xs = t(lambda: torch.addmm(-(mean @ V), X, V)); print("addmm rotate: %.3f ms" % ms)
ms, _ = t(lambda: X @ V); print("plain mm: %.3f ms" % ms)
from meld_amd.graph import HipOps
ops = HipOps()
ms, _ = t(lambda: ops.col_stats(Xs)); print("col_stats: %.3f ms" % ms)
ms, _ = t(lambda: ops.principal_frame(X, mean, 13)); print("principal_frame total: %.3f ms" % ms)
