"""Time VertexFrequencyCluster (dense, small N) on the device: python tools/time_vfc.py N"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np
import torch

import meld_amd
from bench import synthetic_cells

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.default_rng(0)
X = rng.normal(size=(N, 8))
labels = np.where(X[:, 0] + rng.normal(size=N) > 0, "expt", "ctrl")
op = meld_amd.MELD(knn=7, verbose=0)
dens = op.fit_transform(X, labels)
lik = meld_amd.utils.normalize_densities(dens)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vfc = meld_amd.VertexFrequencyCluster(n_clusters=4, random_state=0)
    vfc.fit(op.graph)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    vfc.transform(op.sample_indicators["expt"], lik["expt"])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    vfc.predict()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
print("VFC N=%d: fit (8 window squarings + eigh) %.0f ms, transform (9 windows) %.0f ms, predict (PCA + KMeans) %.0f ms" % (
    N, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
