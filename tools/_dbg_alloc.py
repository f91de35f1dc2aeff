import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import meld_amd
from bench import synthetic_cells
N = int(sys.argv[1])
X, labels = synthetic_cells(N, 50, seed=0)
Xd = torch.from_numpy(X).cuda()
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    op = meld_amd.MELD(knn=15, chebyshev_order=30, verbose=0); op.fit(Xd)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    st = torch.cuda.memory_stats()
    print("fit %.1f ms; alloc_retries %d, hipMalloc calls %d, reserved %.1f GB, allocated peak %.1f GB" % (1e3*(t1-t0), st["num_alloc_retries"], st["num_device_alloc"], st["reserved_bytes.all.current"]/1e9, st["allocated_bytes.all.peak"]/1e9), flush=True)
# time the C call alone
from meld_amd import graph as mg
import meld_amd._lib as L
lib = L.get_lib()
orig = lib.meld_knn16_topk
def timed(*a):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(*a)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("   meld_knn16_topk: call returned after %.2f ms, device done after %.2f ms" % (1e3*(t1-t0), 1e3*(t2-t0)), flush=True)
    return r
lib.meld_knn16_topk = timed
op = meld_amd.MELD(knn=15, chebyshev_order=30, verbose=0); op.fit(Xd)
lib.meld_knn16_topk = orig
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    op = meld_amd.MELD(knn=15, beta=60, chebyshev_order=30, verbose=0); op.fit(Xd)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = op.transform(labels)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("fit %.1f ms, transform %.1f ms (lanczos its %s)" % (1e3*(t1-t0), 1e3*(t2-t1), op.graph.lmax_info.get("iterations")), flush=True)
    t0 = time.perf_counter(); op.graph._lmax = None; op.graph.estimate_lmax(); torch.cuda.synchronize(); print("   estimate_lmax alone %.1f ms" % (1e3*(time.perf_counter()-t0)))
