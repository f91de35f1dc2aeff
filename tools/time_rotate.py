"""Where the principal-frame rotation of the search stage spends its time (1M x 50)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
from bench import synthetic_cells
from meld_amd.graph import HipOps, _stream
from meld_amd._lib import ptr, check
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X = torch.from_numpy(synthetic_cells(n, 50, seed=0)[0]).cuda()
mean = X.mean(0)
ops = HipOps(); lib = ops.lib
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out
cov = torch.zeros(50, 50, dtype=torch.float64, device="cuda")
ms, _ = t(lambda: check(lib.meld_cov_sample_f64(ptr(X), n, 50, ptr(mean), max(1, n // 32768), ptr(cov), _stream()), "cov")); print("cov kernel: %.3f ms" % ms)
ms, covh = t(lambda: cov.cpu().numpy()); print("cov D2H: %.3f ms" % ms)
ms, (ev, evec) = t(lambda: np.linalg.eigh(covh, UPLO="U")); print("numpy eigh 50x50: %.3f ms" % ms)
V = torch.from_numpy(np.ascontiguousarray(evec[:, ::-1])).to(X.device)
Ath = np.zeros((50, 64)); Ath[:, :50] = evec[:, ::-1].T
ms, At = t(lambda: torch.from_numpy(Ath).to(X.device)); print("At H2D: %.3f ms" % ms)
out = torch.empty_like(X)
ms, _ = t(lambda: check(lib.meld_rotate_rows_f64(ptr(X), n, 50, ptr(mean), ptr(At), ptr(out), _stream()), "rot")); print("rotate kernel: %.3f ms" % ms)
ref = (X - mean) @ V
print("rotate max abs err vs torch: %.2e" % float((out - ref).abs().max()))
ms, _ = t(lambda: ops.col_stats(out)); print("col_stats: %.3f ms" % ms)
ms, _ = t(lambda: ops.principal_frame(X, mean, 13)); print("principal_frame total: %.3f ms" % ms)
