"""Low-dimensional, dense data: rows that fail certification vs the candidate-list length.
python tools/lowd_ksel.py [N] [d]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import torch
import meld_amd
from bench import synthetic_cells

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 2
X, labels = synthetic_cells(N, d, seed=0)
Xd = torch.from_numpy(X).cuda()
for ksel in (64, 96, 128):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        op = meld_amd.MELD(knn=15, verbose=0).fit(Xd, ksel=ksel)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("ksel %d: fit %.1f ms, rows through the exact sweep %d" % (ksel, 1e3 * dt, op.graph.info["n_flagged_rows"]))
