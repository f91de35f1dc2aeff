"""One-off full parity check at 200k cells (too slow for the test tier): oracle brute-force kNN on all host cores.
python tools/parity_200k.py [N]      (other shapes / options: DIMS=20 KNN=5 SEED=3 OPTS='{"bandwidth_scale": 0.9}')"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np
from scipy import sparse
import meld_amd
from oracle import meld_oracle as mo

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
import json as _json
DIMS, KNN, SEED = int(os.environ.get("DIMS", "50")), int(os.environ.get("KNN", "15")), int(os.environ.get("SEED", "0"))
OPTS = _json.loads(os.environ.get("OPTS", "{}"))  # graph keywords handed to both sides (bandwidth_scale, knn_max, kernel_symm ...)
if (DIMS, KNN, SEED) != (50, 15, 0) or OPTS:
    print("shape: N=%d d=%d knn=%d seed=%d options=%s" % (N, DIMS, KNN, SEED, OPTS))
X, labels = mo.synthetic_cells(N, n_dims=DIMS, seed=SEED)
t0 = time.perf_counter()
G = mo.build_graph(X, knn=KNN, algorithm="brute", n_jobs=-1, **OPTS)
samples, ind = mo.sample_indicators(labels)
dens = mo.meld_filter(ind, G, beta=60, chebyshev_order=30)
t_oracle = time.perf_counter() - t0
print("oracle: %.1f s" % t_oracle)
op = meld_amd.MELD(knn=KNN, beta=60, chebyshev_order=30, lmax=G.lmax, verbose=0, **OPTS)
out = op.fit_transform(X, labels)
A, B = sparse.csr_matrix(op.graph.W), sparse.csr_matrix(G.W)
A.sort_indices(); B.sort_indices()
print("nnz equal:", A.nnz == B.nnz, " pattern equal:", np.array_equal(A.indices, B.indices) and np.array_equal(A.indptr, B.indptr))
print("max rel weight diff: %.3e" % (np.abs(A.data - B.data).max() / np.abs(B.data).max()))
print("max rel density diff: %.3e" % (np.abs(out.values - dens).max() / np.abs(dens).max()))
print("blocks computed: %.3f, rows re-searched %d, swept %d" % ((op.graph.info.get("wave_tiles_done") or 0) / ((N / 64) ** 2), op.graph.info["n_researched_rows"], op.graph.info["n_flagged_rows"]))
if os.environ.get("MELD_CPU_FULL_JSON"):  # the MEASURED CPU number at this size for bench.py's cpu_baseline.measured_full_size
    import json
    path = os.environ["MELD_CPU_FULL_JSON"]
    try:
        rec = json.load(open(path))
    except Exception:
        rec = {}
    rec[str(N)] = {"cells": N, "seconds": t_oracle, "cells_per_s": N / t_oracle, "cores": os.cpu_count(), "kind": "port",
                   "what": "whole oracle fit_transform (sklearn brute-force kNN on all host cores, n_jobs=-1, + scipy Chebyshev), measured at full size",
                   "commit": os.environ.get("MELD_COMMIT", "unknown"),
                   "pattern_equal": bool(A.nnz == B.nnz and np.array_equal(A.indices, B.indices) and np.array_equal(A.indptr, B.indptr)),
                   "max_rel_density_diff": float(np.abs(out.values - dens).max() / np.abs(dens).max())}
    json.dump(rec, open(path, "w"), indent=1, sort_keys=True)
