"""Where the HOST time of one fit_transform goes (cProfile, 1M cells): python tools/host_profile_step.py [N]"""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import torch, meld_amd
from bench import synthetic_cells
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X, labels = synthetic_cells(N, 50, seed=0)
Xd = torch.from_numpy(X).cuda()
for _ in range(3):
    out = meld_amd.MELD(knn=15, beta=60, chebyshev_order=30, verbose=0).fit_transform(Xd, labels)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    out = meld_amd.MELD(knn=15, beta=60, chebyshev_order=30, verbose=0).fit_transform(Xd, labels)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
