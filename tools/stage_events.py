import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
import meld_amd
from meld_amd import graph as mg
from oracle import meld_oracle as mo
n = 1000000
X, labels = mo.synthetic_cells(n, n_dims=50, seed=0)
Xd = torch.from_numpy(X).cuda()
for r in range(2):
    mg.record_events(True)
    G = meld_amd.build_knn_graph(Xd, knn=15, profile=True)
    torch.cuda.synchronize()
    print({k: [round(v, 2) for v in vs] for k, vs in mg.event_times_ms().items()}, G.info["n_researched_rows"], G.info["n_flagged_rows"], {k: round(v*1e3,1) for k,v in G.info["stage_seconds"].items()})
