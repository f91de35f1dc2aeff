"""One fit_transform step under the torch profiler: where the HOST spends its time (the GPU side is in the rocprofv3 traces).
python tools/profile_step.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import torch
import meld_amd
from bench import synthetic_cells
from torch.profiler import profile, ProfilerActivity

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X, labels = synthetic_cells(N, 50, seed=0)
Xd = torch.from_numpy(X).cuda()
for _ in range(3):
    meld_amd.MELD(knn=15, beta=60, chebyshev_order=30, verbose=0).fit_transform(Xd, labels)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    t0 = time.perf_counter()
    meld_amd.MELD(knn=15, beta=60, chebyshev_order=30, verbose=0).fit_transform(Xd, labels)
    torch.cuda.synchronize()
    print("step under the profiler: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=55))
