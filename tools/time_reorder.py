import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meld_amd import reorder
from meld_amd._lib import get_lib, ptr, check
from oracle import meld_oracle as mo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X, _ = mo.synthetic_cells(n, n_dims=50, seed=0)
Xd = torch.from_numpy(X).cuda()
for r in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    p = reorder.locality_permutation(Xd)
    torch.cuda.synchronize(); print("locality_permutation %.1f ms" % (1e3 * (time.perf_counter() - t)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); p = reorder.locality_permutation(Xd); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
