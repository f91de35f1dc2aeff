"""Where the locality permutation spends its time: python tools/time_reorder.py [N] (env MELD_REORDER)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import torch
from meld_amd import reorder as ro
from bench import synthetic_cells

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X, _ = synthetic_cells(N, 50, seed=0)
Xd = torch.from_numpy(X).cuda()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = ro.locality_permutation(Xd)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
print("total %.2f ms" % (tot * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    p = ro.locality_permutation(Xd); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=50))
import hashlib
print("permutation sha1 %s" % hashlib.sha1(p.cpu().numpy().tobytes()).hexdigest()[:16])
