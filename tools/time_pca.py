"""Time the PCA front-end (meld_amd/pca.py) on the GPU: python tools/time_pca.py N G k"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import torch

from meld_amd.pca import pca_project

N, G, k = (int(v) for v in sys.argv[1:4])
gen = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn(N, 16, dtype=torch.float64, device="cuda", generator=gen) @ torch.randn(16, G, dtype=torch.float64, device="cuda", generator=gen)
X += 0.1 * torch.randn(N, G, dtype=torch.float64, device="cuda", generator=gen)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Y = pca_project(X, k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("PCA %d x %d -> %d: %.1f ms (%.2f TFLOP/s fp64 on the covariance GEMM alone)" % (N, G, k, 1e3 * dt, 2.0 * N * G * G / dt / 1e12))
