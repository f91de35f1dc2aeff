"""How local are the columns of W in the locality order?  Fraction of nonzeros with |col - row| below a window.
python tools/col_window.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import torch
import meld_amd
from bench import synthetic_cells

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X, _ = synthetic_cells(N, 50, seed=0)
op = meld_amd.MELD(knn=15, verbose=0).fit(torch.from_numpy(X).cuda())
G = op.graph
rows = torch.repeat_interleave(torch.arange(G.n_rows, device="cuda"), G.rowptr[1:] - G.rowptr[:-1])
dist = (G.col.to(torch.int64) - rows).abs()
print("nnz", G.nnz)
for w in (64, 256, 1024, 2048, 4096, 8192, 16384, 65536):
    print("  |col - row| < %6d : %.3f" % (w, float((dist < w).double().mean())))
# distinct 128-byte lines (8 rows of p=2 fp64) touched per 32-row block vs entries
blk = rows // 32
line = G.col.to(torch.int64) // 8
key = torch.unique(blk * (N // 8 + 1) + line)
print("entries per 32-row block %.0f, distinct 128-B lines per block %.0f" % (G.nnz / (N / 32), key.numel() / (N / 32)))
