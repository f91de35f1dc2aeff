import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import meld_amd
from oracle import meld_oracle as mo
n = 1000000
X, labels = mo.synthetic_cells(n, n_dims=50, seed=0)
Xd = torch.from_numpy(X).cuda()
op = meld_amd.MELD(knn=15, beta=60, chebyshev_order=30)
op.fit(Xd); op.transform(labels)
op2 = meld_amd.MELD(knn=15, beta=60, chebyshev_order=30).fit(op.graph)
op.graph.lmax = None
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter(); out = op2.transform(labels); torch.cuda.synchronize(); dt = time.perf_counter() - t
pr.disable()
print("transform %.1f ms (lanczos its %d)" % (1e3 * dt, op.graph.lmax_info.get("iterations", -1)))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
