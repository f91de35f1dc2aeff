"""Distribution of the SpMM spans inside a filter-bank VertexFrequencyCluster fit at 1M cells: python tools/vfc_spans.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch, meld_amd
from meld_amd import graph as mg
from bench import synthetic_cells
X, labels = synthetic_cells(1000000, 50, seed=0)
op = meld_amd.MELD(knn=15, verbose=0); op.fit_transform(torch.from_numpy(X).cuda(), labels)
G = op.graph
for rep in range(2):
    mg.record_events(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vfc = meld_amd.VertexFrequencyCluster(n_clusters=6, random_state=0, n_init=2); vfc.fit(G)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ev = mg.event_times_ms(); mg.record_events(False)
    t = np.array(ev["vfc_spmm"])
    print("fit %.3f s; spans %d: mean %.3f median %.3f p10 %.3f p90 %.3f max %.3f ms (%s)" % (t1 - t0, len(t), t.mean(), np.median(t), np.percentile(t, 10), np.percentile(t, 90), t.max(), vfc._fb["spmm"]))
