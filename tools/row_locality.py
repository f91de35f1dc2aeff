"""How far (in rows of the device order) a cell's neighbours lie: the share of the nonzeros within a window of rows -- what an XCD's L2
can hold of a wide iterate decides how often the wide recurrence step re-reads it: python tools/row_locality.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import torch, meld_amd
from bench import synthetic_cells
N = 1_000_000
X, _ = synthetic_cells(N, 50, seed=0)
G = meld_amd.MELD(knn=15, verbose=0).fit(torch.from_numpy(X).cuda()).graph
rp = G.rowptr.long(); col = G.col.long()
rows = torch.repeat_interleave(torch.arange(N, device="cuda"), rp[1:] - rp[:-1])
d = (col - rows).abs().float()
qs = torch.tensor([0.25, 0.5, 0.75, 0.9, 0.95, 0.99], device="cuda")
# quantile on a sample
idx = torch.randint(0, d.numel(), (4_000_000,), device="cuda")
print("nnz", d.numel(), "|col-row| quantiles", [int(v) for v in torch.quantile(d[idx], qs)])
for w in (2048, 4096, 8192, 16384, 32768, 65536, 131072):
    print("within", w, "rows: %.3f" % float((d < w).float().mean()))
