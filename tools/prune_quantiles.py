"""How much of the pruned search's work is owed to a few rows per wave?  Final per-row thresholds (radius cut) and
the per-wave bound table: fraction of (wave, tile) pairs live when the wave's bound is the max / a quantile of
its rows' thresholds.   python tools/prune_quantiles.py [N] [knn]"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
from meld_amd._lib import get_lib, ptr, check
from meld_amd.reorder import locality_permutation
from bench import synthetic_cells

lib = get_lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
knn = int(sys.argv[2]) if len(sys.argv) > 2 else 15
X, _ = synthetic_cells(n, 50, seed=0)
Xd = torch.from_numpy(X).cuda()
Xd = Xd.index_select(0, locality_permutation(Xd)).contiguous()
N, d = Xd.shape
st = torch.cuda.current_stream().cuda_stream
TS, BQ = lib.meld_knn16_tile_refs(), lib.meld_knn16_block_queries()
ksel = 64; cap = lib.meld_knn16_row_capacity(ksel)
sums = torch.empty(d, dtype=torch.float64, device="cuda"); check(lib.meld_col_sums_f64(ptr(Xd), N, d, ptr(sums), st)); mean = sums / N
n_tiles = (N + TS - 1) // TS; q_pad = ((N + BQ - 1) // BQ) * BQ
Rt = torch.empty(n_tiles * lib.meld_knn16_tile_bytes(d), dtype=torch.uint8, device="cuda")
Q = torch.empty(q_pad * lib.meld_knn16_query_bytes(d), dtype=torch.uint8, device="cuda"); Qn = torch.empty(q_pad, dtype=torch.float32, device="cuda")
norm2 = torch.empty(N, dtype=torch.float32, device="cuda"); nmax = torch.zeros(1, dtype=torch.float32, device="cuda"); sinfo = torch.empty(4, dtype=torch.float32, device="cuda")
check(lib.meld_knn16_prepare(ptr(Xd), N, d, ptr(mean), 0, N, ptr(Rt), ptr(Q), ptr(Qn), ptr(norm2), ptr(nmax), ptr(sinfo), st))
tmpb = torch.empty(lib.meld_knn16_bounds_temp_bytes(N, d, N), dtype=torch.uint8, device="cuda")
lb2 = torch.empty(lib.meld_knn16_bounds_bytes(N, N), dtype=torch.uint8, device="cuda")
check(lib.meld_knn16_bounds(ptr(Xd), N, d, ptr(mean), ptr(sinfo), ptr(nmax), ptr(Rt), 0, N, None, None, 1, ptr(tmpb), ptr(lb2), st))
ci = torch.empty(q_pad * cap, dtype=torch.int32, device="cuda"); cd = torch.empty(q_pad * cap, dtype=torch.float32, device="cuda"); cc = torch.empty(q_pad, dtype=torch.int32, device="cuda")
cthr = torch.full((q_pad,), float("inf"), dtype=torch.float32, device="cuda"); done = torch.zeros(1, dtype=torch.int64, device="cuda")
rf = (-math.log(1e-4)) ** (1 / 40)
check(lib.meld_knn16_topk(ptr(Q), ptr(Qn), ptr(Rt), ptr(sinfo), N, d, N, ksel, 1, 1, ptr(lb2), ptr(nmax), 0, None, knn, rf, ptr(ci), ptr(cd), ptr(cc), ptr(cthr), ptr(done), None, st))
torch.cuda.synchronize()
s2 = float(sinfo[0]) ** 2
n_w = q_pad // 64
print("kernel computed %.3f of the (wave, tile) pairs" % (float(done) / (n_w * n_tiles)))
thr = (cthr * s2).view(n_w, 64)                      # final thresholds, scaled units
margin = float(lib.meld_knn16_error_coef(1, d)) * float(nmax) * s2
L = lb2.view(torch.float16).view(n_w, n_tiles)
srt = torch.sort(thr, dim=1).values
for name, col in (("max", 63), ("2nd largest", 62), ("4th largest", 60), ("p90 (7th largest)", 57), ("median", 32)):
    b = srt[:, col] + margin
    live = 0
    for w0 in range(0, n_w, 2048):
        live += int((L[w0:w0 + 2048].float() <= b[w0:w0 + 2048, None]).sum())
    print("wave bound = %-18s: %.3f of the pairs live (final thresholds)" % (name, live / (n_w * n_tiles)))
print("threshold spread within a wave: median of max/median = %.2f, p90 = %.2f" % (float((srt[:, 63] / srt[:, 32]).median()), float(torch.quantile((srt[:, 63] / srt[:, 32])[:100000], 0.9))))
