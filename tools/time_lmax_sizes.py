"""The lmax estimate on the benchmark graphs: wall time against iterations x kernel time (what the host's convergence checks cost):
python tools/time_lmax_sizes.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import torch, meld_amd
from bench import synthetic_cells
for N in [int(a) for a in sys.argv[1:]] or [1_000_000, 500_000]:
    X, _ = synthetic_cells(N, 50, seed=0)
    G = meld_amd.MELD(knn=15, verbose=0).fit(torch.from_numpy(X).cuda()).graph
    G.ops.pt_layout(G)
    for rep in range(4):
        G._lmax = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        G.estimate_lmax()
        torch.cuda.synchronize(); t1 = time.perf_counter()
    info = G.lmax_info
    print("N=%d: lmax %.3f ms, %d iterations (%.1f us each), residual %.2e, lmax %.12g  %s" % (N, 1e3 * (t1 - t0), info["iterations"], 1e6 * (t1 - t0) / info["iterations"], info["residual"], G.lmax, {k: v for k, v in os.environ.items() if k.startswith("MELD_L")}))
