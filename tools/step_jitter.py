"""Step-to-step spread of fit_transform at the benchmark size: python tools/step_jitter.py [steps]  (env toggles of the product)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
import meld_amd
from bench import synthetic_cells

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
X, labels = synthetic_cells(1_000_000, 50, seed=0)
Xd = torch.from_numpy(X).cuda()
ts = []
for i in range(steps + 2):
    op = meld_amd.MELD(knn=15, beta=60, chebyshev_order=30, verbose=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    op.fit_transform(Xd, labels)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
ts = ts[2:]
print("median %.2f  max %.2f  steps over median + 5 ms: %d of %d  all %s  env %s" % (np.median(ts), max(ts), sum(t > np.median(ts) + 5 for t in ts), len(ts),
      [round(t, 1) for t in ts], {k: v for k, v in os.environ.items() if k.startswith("MELD_")}))
