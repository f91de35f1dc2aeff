"""PCIe-inclusive step: fit_transform with X as a host NumPy array (python tools/time_host_input.py [N])"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np
import torch

import meld_amd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_cells

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X, labels = synthetic_cells(N)
Xd = torch.from_numpy(X).cuda()
for name, inp in (("device tensor", Xd), ("host ndarray", X), ("host ndarray", X)):
    ts = []
    for r in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        meld_amd.MELD(knn=15, beta=60, chebyshev_order=30, verbose=0).fit_transform(inp, labels)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print("%-14s: %.1f ms/step (best of 3; %.2f M cells/s)" % (name, 1e3 * min(ts), N / min(ts) / 1e6))
