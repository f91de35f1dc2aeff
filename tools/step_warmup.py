"""How many steps until fit_transform reaches its steady time, and what the allocator does meanwhile: python tools/step_warmup.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
import meld_amd
from bench import synthetic_cells

X, labels = synthetic_cells(1_000_000, 50, seed=0)
Xd = torch.from_numpy(X).cuda()
prev = 0
for i in range(12):
    op = meld_amd.MELD(knn=15, beta=60, chebyshev_order=30, verbose=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    op.fit_transform(Xd, labels)
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0)
    st = torch.cuda.memory_stats()
    segs = st["num_device_alloc"]
    print("step %2d: %8.2f ms   device allocations so far %d (+%d)  reserved %.2f GB  retries %d" % (i, ms, segs, segs - prev, st["reserved_bytes.all.current"] / 1e9, st["num_alloc_retries"]))
    prev = segs
    del op
