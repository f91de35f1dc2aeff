"""Run only the graph-build stage (for profiling the search kernel): python tools/knn_only.py N [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import meld_amd
from meld_amd import graph as mg
from meld_amd.graph import HipOps
from oracle import meld_oracle as mo
reps = 2
for n in [int(a) for a in sys.argv[1].split(",")]:
    X, _ = mo.synthetic_cells(n, n_dims=50, seed=0)
    Xd = torch.from_numpy(X).cuda()
    if os.environ.get('ZERO'):
        Xd = torch.zeros_like(Xd); Xd[0, 0] = 1.0
    ops = HipOps()
    for r in range(reps):
        mg.record_events(True)
        keys, vals, bw, info = ops.directed_kernel_coo(Xd, 0, n, 15, 40, 1e-4, 64)
        torch.cuda.synchronize()
        ms = mg.event_times_ms()["knn_topk"][0]
    ideal = (n / 32.0) ** 2 * 12 * 32 / 1024 / 2.4e9 * 1e3
    print("N=%d knn_topk %.2f ms  ideal@2.4GHz %.2f ms  util %.1f%%  flagged %d  ns/pair %.4f" % (n, ms, ideal, 100 * ideal / ms, info["n_flagged_rows"], ms * 1e6 / n / n))
