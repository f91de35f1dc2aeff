"""How much of the pruning-table product could a sphere-sphere pre-test skip?  For every (cell tile, group of G reference tiles):
dead if |c_t - C_g| - rho_t - R_g > max seed radius of the tile's cells.   python tools/sim_coarse_bounds.py [N]"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
from meld_amd._lib import get_lib, ptr, check
from meld_amd.reorder import locality_permutation
from bench import synthetic_cells

lib = get_lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X, _ = synthetic_cells(n, 50, seed=0)
Xd = torch.from_numpy(X).cuda()
Xd = Xd.index_select(0, locality_permutation(Xd)).contiguous()
N, d = Xd.shape
st = torch.cuda.current_stream().cuda_stream
TS, BQ = 64, lib.meld_knn16_block_queries()
sums = torch.empty(d, dtype=torch.float64, device="cuda"); check(lib.meld_col_sums_f64(ptr(Xd), N, d, ptr(sums), st)); mean = sums / N
n_tiles = (N + TS - 1) // TS; q_pad = ((N + BQ - 1) // BQ) * BQ
Rt = torch.empty(n_tiles * lib.meld_knn16_tile_bytes(d), dtype=torch.uint8, device="cuda")
Q = torch.empty(q_pad * lib.meld_knn16_query_bytes(d), dtype=torch.uint8, device="cuda"); Qn = torch.empty(q_pad, dtype=torch.float32, device="cuda")
norm2 = torch.empty(N, dtype=torch.float32, device="cuda"); nmax = torch.zeros(1, dtype=torch.float32, device="cuda"); sinfo = torch.empty(4, dtype=torch.float32, device="cuda")
check(lib.meld_knn16_prepare(ptr(Xd), N, d, ptr(mean), 0, N, ptr(Rt), ptr(Q), ptr(Qn), ptr(norm2), ptr(nmax), ptr(sinfo), st))
seed = torch.empty(q_pad, dtype=torch.float32, device="cuda")
check(lib.meld_knn16_seed_thresholds_mfma(ptr(Q), ptr(Qn), ptr(Rt), ptr(sinfo), ptr(nmax), N, d, 0, N, 15, (-math.log(1e-4)) ** (1 / 40), 1, 0, ptr(seed), st))
s = float(sinfo[0])
rad_q = torch.sqrt(seed[:N].double()) / s                      # seed radius of every cell, input units
nt = N // TS
Xt = Xd[: nt * TS].view(nt, TS, d)
c = Xt.mean(1)                                                 # tile centres (plain centroids: the product's are a little tighter)
rho = (Xt - c[:, None, :]).norm(dim=2).max(1).values
smax = rad_q[: nt * TS].view(nt, TS).max(1).values             # largest seed radius among the tile's cells
print("tiles %d: radius median %.2f, seed radius median %.2f (max per tile median %.2f)" % (nt, float(rho.median()), float(rad_q.median()), float(smax.median())))
for G in (4, 16, 64):
    ng = nt // G
    cg = c[: ng * G].view(ng, G, d)
    C = cg.mean(1)
    R = ((cg - C[:, None, :]).norm(dim=2) + rho[: ng * G].view(ng, G)).max(1).values
    D = torch.cdist(c, C)                                      # [nt, ng]
    dead = (D - rho[:, None] - R[None, :]) > smax[:, None]
    print("groups of %2d tiles: radius median %.2f; (cell tile, group) pairs dead by the sphere-sphere test: %.3f" % (G, float(R.median()), float(dead.double().mean())))
# the fine per-tile sphere-sphere test for comparison
D = torch.cdist(c, c)
print("tile x tile sphere-sphere: dead %.3f" % float(((D - rho[:, None] - rho[None, :]) > smax[:, None]).double().mean()))
