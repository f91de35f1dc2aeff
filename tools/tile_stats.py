"""How do the nonzeros of W fall into (row block x column panel) tiles in the locality order?
Decides the shape of the panel-staged recurrence kernel.

python tools/tile_stats.py [N]

For row blocks of R rows and column panels of C columns: tiles touched per row block, share of the
nonzeros in tiles of at least T entries ("dense" tiles, worth staging the panel of the iterate in
LDS), dense tiles per block, distinct columns per block.  Optionally dumps the subgraph of the first
65536 rows (columns clipped to it) for offline ordering experiments."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np
import torch
import meld_amd
from bench import synthetic_cells

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X, _ = synthetic_cells(N, 50, seed=0)
op = meld_amd.MELD(knn=15, verbose=0).fit(torch.from_numpy(X).cuda())
G = op.graph
rowptr = G.rowptr.cpu().numpy()
col = G.col.cpu().numpy().astype(np.int64)
nnz = col.size
rows = np.repeat(np.arange(G.n_rows, dtype=np.int64), np.diff(rowptr))
print("N", N, "nnz", nnz, "mean row", nnz / N, "max row", int(np.diff(rowptr).max()))

for w in (1024, 4096, 16384, 65536):
    print("  |col - row| < %6d : %.3f" % (w, float((np.abs(col - rows) < w).mean())))

for R in (1024, 2048, 4096):
    nb = (N + R - 1) // R
    for C in (1024, 2048, 4096):
        npan = (N + C - 1) // C
        tile = (rows // R) * npan + col // C
        tid, cnt = np.unique(tile, return_counts=True)
        tb = tid // npan
        tiles_per_block = np.bincount(tb, minlength=nb)
        line = "R=%4d C=%4d: tiles/block mean %.1f max %d |" % (R, C, tiles_per_block.mean(), tiles_per_block.max())
        for T in (32, 64, 128, 256, 512):
            dense = cnt >= T
            share = cnt[dense].sum() / nnz
            per_block = np.bincount(tb[dense], minlength=nb)
            line += " T>=%d: %.3f nnz, %.1f (max %d) tiles/blk |" % (T, share, per_block.mean(), per_block.max())
        print(line, flush=True)
    key = np.unique((rows // R) * N + col)
    print("R=%4d: distinct columns per block mean %.0f (entries per block %.0f)" % (R, key.size / nb, nnz / nb), flush=True)

os.makedirs("gpurun_out/tiles", exist_ok=True)
M = 131072
sel = (rows < M) & (col < M)
rp = np.zeros(M + 1, dtype=np.int64)
np.add.at(rp, rows[sel] + 1, 1)
rp = np.cumsum(rp)
np.savez_compressed("gpurun_out/tiles/sub_%d.npz" % M, rowptr=rp, col=col[sel].astype(np.int32),
                    X=X[op.graph.perm.cpu().numpy()[:M]].astype(np.float32) if hasattr(op.graph, "perm") and op.graph.perm is not None else np.zeros(1))
print("kept", int(sel.sum()), "of", int((rows < M).sum()), "entries of the first", M, "rows")
