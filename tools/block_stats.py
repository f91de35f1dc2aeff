"""Structure of W in the locality order, as the recurrence kernel's row blocks see it (1M benchmark graph):
python tools/block_stats.py graph.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import torch

d = torch.load(sys.argv[1])
rowptr, col = d["rowptr"].cuda(), d["col"].cuda().to(torch.int64)
N = int(d["N"]); nnz = col.numel()
rows = torch.repeat_interleave(torch.arange(N, device="cuda"), rowptr[1:] - rowptr[:-1])
print("N %d nnz %d" % (N, nnz))
for R in (977, 1953, 3906, 7812, 15625, 31250, 125000):
    br, bc = rows // R, col // R
    inb = float((br == bc).double().mean())
    adj = float(((br - bc).abs() <= 1).double().mean())
    nblk = (N + R - 1) // R
    key = torch.unique(br * N + col)
    ndist = key.numel() / nblk
    # runs of consecutive columns within a block's sorted distinct list; distinct 64-B / 128-B lines (p = 2 fp64 rows)
    kb, kc = key // N, key % N
    newrun = torch.ones_like(key, dtype=torch.bool)
    newrun[1:] = (kb[1:] != kb[:-1]) | (kc[1:] != kc[:-1] + 1)
    nruns = int(newrun.sum()) / nblk
    l64 = torch.unique(kb * N + kc // 4).numel() / nblk
    l128 = torch.unique(kb * N + kc // 8).numel() / nblk
    print("rows/block %6d: in-block entries %.3f (+-1 block %.3f); distinct cols/block %.0f (%.2f x rows; %.2f entries per col); "
          "runs %.0f (mean len %.2f); 64-B lines %.0f, 128-B lines %.0f" % (R, inb, adj, ndist, ndist / R, nnz / nblk / ndist, nruns, ndist / nruns, l64, l128))
# out-of-block columns only: how many distinct, how often used
R = 3906
br, bc = rows // R, col // R
out = br != bc
key, cnt = torch.unique(br[out] * N + col[out], return_counts=True)
print("R=3906: out-of-block entries %.3f of nnz, distinct out-of-block cols per block %.0f, uses per col mean %.2f; cols used once %.3f"
      % (float(out.double().mean()), key.numel() / (N / R), float(cnt.double().mean()), float((cnt == 1).double().mean())))
deg = (rowptr[1:] - rowptr[:-1])
print("row degree: mean %.1f max %d; p50 %d p99 %d" % (float(deg.double().mean()), int(deg.max()), int(deg.double().quantile(0.5)), int(deg.double().quantile(0.99))))
