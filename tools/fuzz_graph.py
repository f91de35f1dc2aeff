"""One-off fuzz of the graph builder against the oracle on small awkward inputs.  python tools/fuzz_graph.py [n_cases] [seed]"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
from scipy import sparse
import meld_amd
from oracle import meld_oracle as mo

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
kinds = ["gauss", "clusters", "duplicates", "grid", "multiscale", "line", "heavy"]
bad = 0
for c in range(n_cases):
    kind = kinds[rng.integers(len(kinds))]
    N = int(rng.integers(int(os.environ.get("FUZZ_N_MIN", "300")), int(os.environ.get("FUZZ_N_MAX", "20000")))); d = int(rng.integers(int(os.environ.get("FUZZ_D_MIN", "1")), int(os.environ.get("FUZZ_D_MAX", "60")))); knn = int(rng.integers(int(os.environ.get("FUZZ_KNN_MIN", "1")), int(os.environ.get("FUZZ_KNN_MAX", "25"))))
    decay = float(rng.choice([2, 10, 40, 100])); thresh = float(rng.choice([1e-2, 1e-4, 1e-6])); aniso = float(rng.choice([0, 1]))
    if kind == "gauss": X = rng.normal(size=(N, d))
    elif kind == "clusters": X = rng.normal(size=(N, d)) * 0.3 + rng.normal(size=(8, d))[rng.integers(0, 8, N)] * 4
    elif kind == "duplicates":
        base = rng.normal(size=(max(N // 3, 10), d)); X = base[rng.integers(0, base.shape[0], N)]
    elif kind == "grid": X = rng.integers(0, 6, size=(N, d)).astype(float)
    elif kind == "multiscale": X = np.concatenate([rng.normal(size=(N // 2, d)) * 1e-3, rng.normal(size=(N - N // 2, d)) * 3 + 20])
    elif kind == "line": X = np.outer(np.linspace(0, 1, N), rng.normal(size=d)) + 1e-4 * rng.normal(size=(N, d))
    else: X = rng.standard_t(2.0, size=(N, d))
    knn = min(knn, N - 2)
    # graphtools' bandwidth options (FUZZ_OPTIONS=1; drawn after everything else, so the plain cases keep their streams)
    extra = {}
    if os.environ.get("FUZZ_OPTIONS"):
        o = rng.integers(0, 5)
        if o == 1: extra = dict(bandwidth_scale=float(rng.choice([0.5, 0.8, 0.95, 1.2, 2.0])))
        elif o == 2: extra = dict(bandwidth=float(np.abs(X).mean() * rng.choice([0.05, 0.3, 1.0])) + 1e-9)
        elif o == 3: extra = dict(knn_max=int(knn + rng.integers(0, 20)))
        elif o == 4: extra = dict(knn_max=int(knn + rng.integers(0, 20)), bandwidth_scale=float(rng.choice([0.8, 1.3])))
        if "knn_max" in extra and kind in ("grid", "duplicates"):
            # (which of several cells at exactly the cut distance a row keeps is the neighbour search's tie order -- sklearn's
            # partition on one side, (distance, index) on the other: not comparable on data made of ties)
            extra.pop("knn_max")
    tag = "%-10s N=%5d d=%2d knn=%2d decay=%g thresh=%g a=%g %s" % (kind, N, d, knn, decay, thresh, aniso, extra or "")
    if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) != c:  # (the generator has been advanced as the full run does)
        continue
    if os.environ.get("FUZZ_NO_ORACLE"):  # product only: does it build, is W symmetric and finite (hundreds of cases a minute)
        try:
            DG = meld_amd.build_knn_graph(torch.from_numpy(np.ascontiguousarray(X)).cuda(), knn=knn, decay=decay, thresh=thresh, anisotropy=aniso, **extra)
            A = sparse.csr_matrix(DG.W)
            asym = abs(A - A.T).max() if A.nnz else 0.0
            okv = np.isfinite(A.data).all() and asym <= 1e-15 * max(abs(A.data).max(), 1e-300) * 4
            bad += not okv
            print("ok  " if okv else "BAD ", tag, "nnz %d asym %.1e flagged %d rebw %d" % (A.nnz, asym, DG.info["n_flagged_rows"], DG.info["n_rows_bandwidth_recomputed"]), flush=True)
        except Exception as e:
            if isinstance(e, ValueError) and "no off-diagonal" in str(e) and ("bandwidth" in extra or "bandwidth_scale" in extra):
                print("ok  ", tag, "no edges (bandwidth option)", flush=True)
                continue
            bad += 1
            print("EXC ", tag, type(e).__name__, str(e)[:120], flush=True)
        continue
    try:
        G = mo.build_graph(X, knn=knn, decay=decay, thresh=thresh, anisotropy=aniso, algorithm="kd_tree" if d <= 20 else "ball_tree", **extra)
        try:
            DG = meld_amd.build_knn_graph(torch.from_numpy(np.ascontiguousarray(X)).cuda(), knn=knn, decay=decay, thresh=thresh, anisotropy=aniso, **extra)
        except ValueError as e:
            # a bandwidth so small that no cell has a neighbour inside its radius: the oracle's graph has no edge, the product says so
            if "no off-diagonal" in str(e) and sparse.csr_matrix(G.W).nnz == 0:
                print("ok  ", tag, "no edges on either side", flush=True)
                continue
            raise
        A, B = sparse.csr_matrix(DG.W), sparse.csr_matrix(G.W)
        A.sort_indices(); B.sort_indices()
        if A.nnz != B.nnz or not np.array_equal(A.indices, B.indices):
            D = abs(A - B); print("MISMATCH pattern", tag, A.nnz, B.nnz, "max|diff|", D.max()); bad += 1; continue
        err = np.abs(A.data - B.data).max() / max(np.abs(B.data).max(), 1e-300)
        flag = "" if err < 1e-9 else "  <-- VALUES"
        bad += err >= 1e-9
        print("ok  ", tag, "nnz %d err %.1e" % (A.nnz, err), flag, flush=True)
    except Exception as e:
        bad += 1
        print("EXC ", tag, type(e).__name__, str(e)[:120], flush=True)
print("bad:", bad)
