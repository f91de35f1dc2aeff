"""Run only the graph-build stage (for profiling the search kernel)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import meld_amd
from meld_amd.graph import HipOps
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rng = np.random.default_rng(0)
from oracle import meld_oracle as mo
X, _ = mo.synthetic_cells(n, n_dims=50, seed=0)
Xd = torch.from_numpy(X).cuda()
ops = HipOps()
for r in range(reps):
    torch.cuda.synchronize(); t = time.perf_counter()
    keys, vals, bw, info = ops.directed_kernel_coo(Xd, 0, n, 15, 40, 1e-4, 64)
    torch.cuda.synchronize(); print("rep", r, time.perf_counter() - t, info)
