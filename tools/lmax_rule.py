"""Stop rules of the lmax estimate, on recorded Lanczos runs: for every prefix k of ONE long run of the device recurrence
(alphas / betas read back once) the top Ritz value, its relative residual (today's rule: <= 1e-3), the eigenvalue-error estimate
residual^2 / gap (gap = theta_1 - theta_2 of the k x k tridiagonal) and the TRUE error against the converged value.
python tools/lmax_rule.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
from scipy.linalg import eigh_tridiagonal
import meld_amd
from meld_amd import filter as mf
from bench import synthetic_cells

sizes = [int(a) for a in sys.argv[1:]] or [1_000_000, 500_000, 200_000]
KMAX = 160
for N in sizes:
    X, _ = synthetic_cells(N, 50, seed=0)
    G = meld_amd.MELD(knn=15, verbose=0).fit(torch.from_numpy(X).cuda()).graph
    ops = mf._ops_of(G)
    dev, n = G.val.device, G.N
    idx = torch.arange(n, dtype=torch.float64, device=dev)
    u = torch.frac(torch.sin(idx * 12.9898 + 1.0) * 43758.5453) - 0.5
    V = torch.zeros(3, n, dtype=torch.float64, device=dev)
    V[1].copy_(u)
    state = torch.zeros(8, dtype=torch.float64, device=dev)
    state[0] = state[3] = 1.0 / torch.linalg.vector_norm(V[1])
    ab = torch.zeros(2, KMAX, dtype=torch.float64, device=dev)
    scratch = torch.zeros(8 * ops.dot_slots(), dtype=torch.float64, device=dev)
    ops.lanczos_steps(G, V, state, ab[0], ab[1], 0, KMAX, scratch, None)
    a, b = ab.cpu().numpy()
    ev = eigh_tridiagonal(a, b[:-1], select="i", select_range=(KMAX - 1, KMAX - 1))[0]
    ref = float(ev[0])
    print("N = %d: converged lambda_max %.12g (k = %d)" % (N, ref, KMAX))
    print("   k   rel.residual   est = res^2/gap (rel)   TRUE rel.err   gap/theta   est/true")
    first = {}
    for k in range(10, 121, 5):
        w, z = eigh_tridiagonal(a[:k], b[: k - 1], select="i", select_range=(k - 2, k - 1))
        th1, th2 = float(w[1]), float(w[0])
        res = abs(b[k - 1] * z[-1, 1])
        gap = max(th1 - th2, 1e-300)
        est = res * res / gap / th1
        true = abs(th1 - ref) / ref
        print("  %3d   %.3e      %.3e               %.3e     %.2e    %.2f" % (k, res / th1, est, true, gap / th1, est / max(true, 1e-300)))
        for name, hit in (("resid<=1e-3", res / th1 <= 1e-3), ("est<=1e-5", est <= 1e-5), ("4est<=1e-5", 4 * est <= 1e-5), ("est<=3e-6", est <= 3e-6),
                          ("resid<=2e-3", res / th1 <= 2e-3), ("resid<=3e-3", res / th1 <= 3e-3)):
            if hit and name not in first:
                first[name] = (k, true)
    print("  first stop per rule (k, true rel.err): " + "; ".join("%s: k=%d err=%.1e" % (nm, kv[0], kv[1]) for nm, kv in first.items()))
