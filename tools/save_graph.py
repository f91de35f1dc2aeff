"""Build the benchmark graph once and keep its CSR on disk for kernel experiments:
python tools/save_graph.py [N] [out.pt]   (then: python tools/spmm_time.py out.pt)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import torch
import meld_amd
from bench import synthetic_cells

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/graph_%d.pt" % n
X, _ = synthetic_cells(n, 50, seed=0)
G = meld_amd.build_knn_graph(torch.from_numpy(X).cuda(), knn=15)
torch.save(dict(rowptr=G.rowptr.cpu(), col=G.col.cpu(), val=G.val.cpu(), dw=G.dw_dev.cpu(), N=G.N), out)
print("saved", out, "N", G.N, "nnz", G.nnz)
