#!/usr/bin/env python
"""Re-derive the golden fixtures G1-G7 from the REAL reference stack and diff them against
tests/golden/*.npz (last sentence).

    python tools/regen_golden_from_reference.py [--reference /root/reference] [--write DIR]

The fixtures under tests/golden/ are outputs of the in-repo oracle (oracle/meld_oracle.py): the
reference package ``meld`` cannot be imported in the build container because its dependencies
graphtools (>= 1.5.0) and pygsp (0.5.1) are not installed and there is no network.  This script is
the route to pinning the oracle's hot path (kNN kernel, lmax, Chebyshev coefficients and recurrence)
on the reference itself: on a machine that has ``graphtools`` + ``pygsp`` (+ ``scprep``,
``tasklogger``; ``pip install meld`` pulls them) it imports the reference package from
``--reference`` (default /root/reference, or $MELD_REFERENCE_PATH), runs every golden case through
``meld.MELD`` / ``graphtools.Graph`` / ``pygsp`` with the fixture's stored lmax injected the way pygsp
allows (``G._lmax``; ``estimate_lmax`` is then a no-op), and reports the largest deviation of

    W (pattern and values), dw, kernel bandwidths, Chebyshev coefficients, sample densities

from the committed fixture.  Exit status: 0 all within tolerance, 1 a deviation, 77 dependencies
missing (nothing compared).  ``--write DIR`` additionally stores the reference-derived vectors as
``DIR/<fixture>.npz`` (same keys) so they can replace the oracle-derived ones.

Nothing here is used by the product or by the GPU tests; tests/test_oracle.py calls ``main()`` and
skips on status 77.
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TOL_W = 1e-9  # relative, on the weights (fp64 distances by two different routes)
TOL_DENS = 1e-5  # north-star tolerance on the densities (relative to the column maximum)
SKIP = 77


def _import_reference(path):
    """(meld, graphtools, pygsp) or None when the stack is not importable."""
    for mod in ("graphtools", "pygsp"):
        try:
            importlib.import_module(mod)
        except Exception as e:  # noqa: BLE001
            print("[regen_golden] {} is not importable ({}): nothing to compare".format(mod, e))
            return None
    if path and os.path.isdir(path) and path not in sys.path:
        sys.path.insert(0, path)
    try:
        meld = importlib.import_module("meld")
    except Exception as e:  # noqa: BLE001
        print("[regen_golden] the reference package `meld` is not importable from {} ({})".format(path, e))
        return None
    import graphtools
    import pygsp

    return meld, graphtools, pygsp


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _run_case(meld, pygsp, X, labels, lmax, meld_kwargs, order):
    """One fit_transform through the real stack with the fixture's lmax injected."""
    op = meld.MELD(verbose=0, **meld_kwargs)
    op.fit(X)
    G = op.graph
    if lmax is not None:
        G._lmax = float(lmax)  # pygsp: estimate_lmax() is a no-op once set
    dens = op.transform(labels)
    out = dict(samples=np.asarray(dens.columns), dens=dens.values, nnz=np.int64(G.W.nnz),
               dw=np.asarray(G.dw), lmax=np.float64(G.lmax))
    W = G.W.tocsr()
    W.sort_indices()
    out.update(rowptr=W.indptr.astype(np.int64), W_data=W.data, W_indices=W.indices.astype(np.int32))
    if order is not None and meld_kwargs.get("solver", "chebyshev") == "chebyshev":
        from pygsp.filters import approximations

        h = pygsp.filters.Filter(G, lambda x: np.exp(-meld_kwargs.get("beta", 60) * np.abs(x / G.lmax)))
        out["coeffs"] = np.asarray(approximations.compute_cheby_coeff(h, m=order)).ravel()
    return out


def _compare(name, got, ref, report):
    worst = 0.0
    for key in ("dens", "dw", "coeffs", "W_data", "bandwidth"):
        if key in got and key in ref.files:
            e = _rel(got[key], ref[key])
            tol = TOL_DENS if key == "dens" else TOL_W
            report.append((name, key, e, tol, e <= tol))
            worst = max(worst, e / tol)
    for key in ("nnz", "rowptr", "W_indices", "samples"):
        if key in got and key in ref.files:
            same = np.array_equal(np.asarray(got[key]), np.asarray(ref[key]))
            report.append((name, key, 0.0 if same else np.inf, 0.0, bool(same)))
            worst = max(worst, 0.0 if same else np.inf)
    return worst


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("MELD_REFERENCE_PATH", "/root/reference"))
    ap.add_argument("--write", default=None, help="directory for the reference-derived fixtures")
    args = ap.parse_args(argv)
    stack = _import_reference(args.reference)
    if stack is None:
        return SKIP
    meld, graphtools, pygsp = stack
    sys.path.insert(0, ROOT)
    from tests.golden import make_golden as mg  # input generators only (seeded data of each fixture)
    from bench import synthetic_cells

    def load(f):
        return np.load(os.path.join(GOLD, f), allow_pickle=False)

    report = []
    results = {}

    # G1: the reference's own known-answer configuration (test/test_meld.py:43-81), exact solver
    g = load("g1_exact_1000x2.npz")
    for filt in ("heat", "laplacian"):
        r = _run_case(meld, pygsp, g["data"], g["labels"], None,
                      dict(knn=20, decay=10, thresh=0, anisotropy=0, filter=filt, solver="exact", sample_normalize=False), None)
        report.append(("g1_" + filt, "dens", _rel(r["dens"], g["dens_" + filt]), TOL_DENS, _rel(r["dens"], g["dens_" + filt]) <= TOL_DENS))
        results["g1_" + filt] = r
    # G2: same data, default sparse Chebyshev path
    g = load("g2_cheby_1000x2.npz")
    results["g2"] = _run_case(meld, pygsp, g["data"], g["labels"], g["lmax"], dict(), 50)
    _compare("g2", results["g2"], g, report)
    # G3: README toy
    g = load("g3_readme_500x100.npz")
    rng = np.random.default_rng(0)
    X = rng.normal(size=(500, 100))
    lab = rng.choice(["treatment", "control"], size=500)
    assert mg.sha(X) == str(g["x_sha"])
    results["g3"] = _run_case(meld, pygsp, X, lab, g["lmax"], dict(), 50)
    _compare("g3", results["g3"], g, report)
    # G4: three labels
    g = load("g4_three_labels_300x2.npz")
    results["g4"] = _run_case(meld, pygsp, g["data"], g["labels"], g["lmax"], dict(), 50)
    _compare("g4", results["g4"], g, report)
    # G5: make_batches(100)
    g = load("g5_batches_600x2.npz")
    results["g5"] = _run_case(meld, pygsp, g["data"], g["labels"], g["lmax"], dict(), 50)
    _compare("g5", results["g5"], g, report)
    # G6: C2-shaped mini
    g = load("g6_c2mini_5000x50.npz")
    X, lab = synthetic_cells(5000, 50, seed=0)
    assert mg.sha(X) == str(g["x_sha"])
    results["g6"] = _run_case(meld, pygsp, X, lab, g["lmax"], dict(knn=15, beta=60, chebyshev_order=30), 30)
    _compare("g6", results["g6"], g, report)

    # G7: the graph keywords MELD forwards to graphtools (bandwidth_scale, bandwidth, knn_max, kernel_symm / theta)
    g = load("g7_graph_options_1000x8.npz")
    X, lab = mg.g7_inputs()
    assert mg.sha(X) == str(g["x_sha"])
    for tag, kw in mg.G7_OPTIONS:
        r = _run_case(meld, pygsp, X, lab, g[tag + "_lmax"], dict(knn=7, beta=60, chebyshev_order=30, **kw), 30)
        results["g7_" + tag] = r
        for key, tol in (("dens", TOL_DENS), ("dw", TOL_W), ("W_data", TOL_W)):
            e = _rel(r[key], g[tag + "_" + key])
            report.append(("g7_" + tag, key, e, tol, e <= tol))
        for key in ("nnz", "rowptr", "W_indices"):
            same = np.array_equal(np.asarray(r[key]), np.asarray(g[tag + "_" + key]))
            report.append(("g7_" + tag, key, 0.0 if same else np.inf, 0.0, bool(same)))

    bad = 0
    for name, key, err, tol, ok in report:
        print("{:12s} {:10s} err {:.3e} (tol {:.0e}) {}".format(name, key, err, tol, "ok" if ok else "DEVIATES"))
        bad += 0 if ok else 1
    if args.write:
        os.makedirs(args.write, exist_ok=True)
        for name, r in results.items():
            np.savez_compressed(os.path.join(args.write, name + "_from_reference.npz"),
                                **{k: v for k, v in r.items() if v is not None})
        print("[regen_golden] reference-derived vectors written to", args.write)
    print("[regen_golden] {} comparisons, {} deviations".format(len(report), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
