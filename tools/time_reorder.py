"""Where the locality permutation spends its time: python tools/time_reorder.py [N] (env MELD_REORDER)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meld_amd import reorder as ro
from bench import synthetic_cells

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X, _ = synthetic_cells(N, 50, seed=0)
Xd = torch.from_numpy(X).cuda()
acc = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    return w
ro._chain_order_batched = timed("chain", ro._chain_order_batched)
ro._split_level = timed("split_level(total)", ro._split_level)
for name in ("argsort", "bincount", "index_select"):
    pass
for rep in range(3):
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = ro.locality_permutation(Xd)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
print("total %.2f ms (with sync overhead)" % (tot * 1e3), {k: round(v * 1e3, 2) for k, v in acc.items()})
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    p = ro.locality_permutation(Xd); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=50))
