"""Timing probe for a narrower index word in the recurrence stream: the step kernels of whatever libmeld_hip.so is in place on the
1M (or N) benchmark graph, self-check bypassed.  Used in round 5 with a timing-only variant of csrc/spmm_tiled.hip (a local patch,
-DPT_IDX16_TIMING: 16-bit index words with the real column and flush bits, a lane-held row; it computed garbage and was not kept --
what it did and what it measured is in profiles/r05_idx16_probe.txt); on the product library it simply times the three step kernels.
python tools/idx16_probe.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
import meld_amd
from meld_amd.graph import HipOps
from bench import synthetic_cells, cheby_bytes_per_step

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X, _ = synthetic_cells(n, 50, seed=0)
G = meld_amd.build_knn_graph(torch.from_numpy(X).cuda(), knn=15)
HipOps._pt_selfcheck[int(torch.cuda.current_device())] = True
G.pt = None
G.ops = HipOps(spmm="tiled")
G.ops.pt_layout(G)


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / reps * 1e3)
    return min(best), float(np.median(best))


for pp in (2, 1):
    x = torch.rand(n, pp, dtype=torch.float64, device="cuda")
    z = torch.rand(n, pp, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    r = torch.zeros_like(x)
    lo, med = timed(lambda: G.ops.cheby_step(G, pp, x, 0, z, y, r, 0.7, -0.2, -1.0, 0.1))
    byts = cheby_bytes_per_step(G.nnz, n, pp)
    print("N=%d tiled p=%d: best %.1f median %.1f us/step  frac(best) %.3f" % (n, pp, lo, med, byts / lo / 1e3 / 8000), flush=True)
ops = G.ops
x1 = torch.rand(n, dtype=torch.float64, device="cuda")
z1 = x1.clone(); y1 = torch.empty_like(x1)
state = torch.zeros(8, dtype=torch.float64, device="cuda"); state[3], state[4] = 0.9, -0.3
dots = torch.zeros(2 * ops.dot_slots(), dtype=torch.float64, device="cuda")
lo, med = timed(lambda: ops.lanczos_spmv(G, x1, z1, y1, state, dots))
print("N=%d lanczos spmv (p=1, fp32 values): best %.1f median %.1f us" % (n, lo, med), flush=True)
