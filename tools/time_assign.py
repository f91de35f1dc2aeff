"""Time meld_assign_nearest alone (level 0 of the ordering: N x 64 centroids): python tools/time_assign.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
from meld_amd._lib import check, get_lib, ptr
from bench import synthetic_cells

lib = get_lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X, _ = synthetic_cells(n, 50, seed=0)
Xd = torch.from_numpy(X).cuda()
st = torch.cuda.current_stream().cuda_stream
for npg in (64, 32):
    cents = Xd[torch.randperm(n, device="cuda")[:npg]].contiguous()
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); check(lib.meld_assign_nearest(ptr(Xd), n, 50, ptr(cents), npg, None, None, ptr(out), st), "assign"); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print("N %d x %d centroids: %.3f ms  (env %s)" % (n, npg, min(ts), {k: v for k, v in os.environ.items() if k.startswith("MELD_ASSIGN")}))
