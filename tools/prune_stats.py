import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
from meld_amd._lib import get_lib, ptr, check
from meld_amd.reorder import locality_permutation
from oracle import meld_oracle as mo
lib = get_lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
X, _ = mo.synthetic_cells(n, n_dims=50, seed=0)
Xd = torch.from_numpy(X).cuda()
Xd = Xd.index_select(0, locality_permutation(Xd))
N, d = Xd.shape
st = torch.cuda.current_stream().cuda_stream
KB, TS, BQ = lib.meld_knn16_kblocks(d), lib.meld_knn16_tile_refs(), lib.meld_knn16_block_queries()
ksel = 64; cap = lib.meld_knn16_row_capacity(ksel)
sums = torch.empty(d, dtype=torch.float64, device="cuda"); check(lib.meld_col_sums_f64(ptr(Xd), N, d, ptr(sums), st)); mean = sums / N
n_tiles = (N + TS - 1) // TS; q_pad = ((N + BQ - 1) // BQ) * BQ
Rt = torch.empty(n_tiles * lib.meld_knn16_tile_bytes(d), dtype=torch.uint8, device="cuda"); Q = torch.empty(q_pad * lib.meld_knn16_query_bytes(d), dtype=torch.uint8, device="cuda"); Qn = torch.empty(q_pad, dtype=torch.float32, device="cuda")
norm2 = torch.empty(N, dtype=torch.float32, device="cuda"); nmax = torch.zeros(1, dtype=torch.float32, device="cuda"); sinfo = torch.empty(4, dtype=torch.float32, device="cuda")
check(lib.meld_knn16_prepare(ptr(Xd), N, d, ptr(mean), 0, N, ptr(Rt), ptr(Q), ptr(Qn), ptr(norm2), ptr(nmax), ptr(sinfo), st))
tmpb = torch.empty(lib.meld_knn16_bounds_temp_bytes(N, d, N), dtype=torch.uint8, device="cuda")
lb2 = torch.empty(lib.meld_knn16_bounds_bytes(N, N) // 4, dtype=torch.float32, device="cuda")
check(lib.meld_knn16_bounds(ptr(Xd), N, d, ptr(mean), ptr(sinfo), 0, N, ptr(tmpb), ptr(lb2), st))
ci = torch.empty(q_pad * cap, dtype=torch.int32, device="cuda"); cd = torch.empty(q_pad * cap, dtype=torch.float32, device="cuda"); cc = torch.empty(q_pad, dtype=torch.int32, device="cuda")
check(lib.meld_knn16_topk(ptr(Q), ptr(Qn), ptr(Rt), ptr(sinfo), N, d, N, ksel, 3, 1, None, None, 0, None, 0, 1.0, ptr(ci), ptr(cd), ptr(cc), None, None, st))
torch.cuda.synchronize()
s = float(sinfo[0]); 
thr = cd.view(q_pad, cap)[:N, ksel - 1] * s * s          # final 64th d2 in scaled units
nq = (N + BQ - 1) // BQ
thr_blk = torch.full((nq * BQ,), 0.0, device="cuda"); thr_blk[:N] = thr
thr_blk = thr_blk.view(nq, BQ).max(1).values
L = lb2.view(nq, n_tiles)
frac_live = (L <= thr_blk[:, None] + 1.5e-5 * float(nmax) * s * s).float().mean().item()
tmp = tmpb.view(torch.float32)
rt = tmp[n_tiles * d: n_tiles * d + n_tiles]; rq = tmp[n_tiles * d + n_tiles + nq * d: n_tiles * d + n_tiles + nq * d + nq]
print("N", N, "scaled: thr_blk median %.4f  tile radius median %.4f  block radius median %.4f  lb2 median %.4f" % (thr_blk.median().item(), rt.median().item(), rq.median().item(), L.median().item()))
print("fraction of (block,tile) pairs that stay live with FINAL thresholds: %.3f" % frac_live)
