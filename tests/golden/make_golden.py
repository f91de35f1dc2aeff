#!/usr/bin/env python
"""Regenerate the golden fixtures of tests/golden/ from the CPU oracle.

    python tests/golden/make_golden.py

The reference package cannot be imported in the build container (graphtools / pygsp absent, no
network), so these vectors come from the in-repo restatement (oracle/meld_oracle.py), which is
pinned to the reference's own known-answer test (G1: sum of the 'treat' density == 532,
reference test/test_meld.py:43-81).  Every fixture stores the injected lmax so that both sides
of a parity test evaluate the same polynomial.  Inputs are small and stored, or regenerated from
a seeded generator whose checksum is stored.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import meld_oracle as mo  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def g1_inputs():
    """Replay of reference test/test_meld.py:47-57 (legacy global RNG, seed 42)."""
    rs = np.random.RandomState(42)
    data = rs.normal(0, 2, (1000, 2))
    x = data[:, 0] - data[:, 0].min()
    x = x / x.max()
    lab = rs.binomial(1, x, 1000)
    labels = np.array(["treat" if v else "ctrl" for v in lab])
    return data, labels


def make_batches(n_pts_per_cluster, seed=0):
    """The fixture of reference test/utils/__init__.py:37-71 (6 Gaussian blobs), seeded."""
    rs = np.random.RandomState(seed)
    centres = [(0, 0), (1, 1), (0, 1), (1, -1), (2, 0), (-2, -1)]
    data = np.concatenate(
        [np.concatenate([rs.normal(x, 0.1, (n_pts_per_cluster, 1)), rs.normal(y, 0.1, (n_pts_per_cluster, 1))], axis=1)
         for x, y in centres], axis=0)
    labels = np.array(["ctrl"] * (3 * n_pts_per_cluster) + ["expt"] * (3 * n_pts_per_cluster))
    return data, labels


def graph_summary(G):
    return dict(nnz=np.int64(G.W.nnz), dw=G.dw, rowptr=G.W.indptr.astype(np.int64), lmax=np.float64(G.lmax))


G7_OPTIONS = [
    ("scale", dict(bandwidth_scale=0.8)),
    ("fixed", dict(bandwidth=2.9)),
    ("knnmax", dict(knn_max=8)),
    ("mnn", dict(kernel_symm="mnn", theta=0.3)),
    ("prod", dict(kernel_symm="*")),
]


def g7_inputs():
    return mo.synthetic_cells(1000, n_dims=8, seed=9)


def make_g7():
    """G7: the graph keywords the reference forwards to graphtools (meld/meld.py:106,117-118) -- bandwidth_scale, bandwidth,
    knn_max, kernel_symm / theta -- on one small data set, knn = 7: per option the symmetrised weights, degrees, lmax and
    densities (Chebyshev order 30).  The keyword names are graphtools' own, so tools/regen_golden_from_reference.py hands the
    same dictionaries to the real stack."""
    X, lab = g7_inputs()
    store = dict(x_sha=sha(X), labels=lab)
    for tag, kw in G7_OPTIONS:
        G = mo.build_graph(X, knn=7, algorithm="brute", **kw)
        samples, ind = mo.sample_indicators(lab)
        dens = mo.meld_filter(ind, G, beta=60, chebyshev_order=30)
        W = G.W.tocsr()
        W.sort_indices()
        store.update({tag + "_dens": dens, tag + "_dw": G.dw, tag + "_lmax": np.float64(G.lmax), tag + "_nnz": np.int64(W.nnz),
                      tag + "_rowptr": W.indptr.astype(np.int64), tag + "_W_indices": W.indices.astype(np.int32), tag + "_W_data": W.data})
    store["samples"] = samples
    np.savez_compressed(os.path.join(HERE, "g7_graph_options_1000x8.npz"), **store)


def main():
    out = {}
    # G1: exact solver on the dense thresh=0 graph (the reference's only known-answer test)
    data, labels = g1_inputs()
    for filt in ("heat", "laplacian"):
        samples, dens = mo.fit_transform(data, labels, knn=20, decay=10, thresh=0, anisotropy=0, filter=filt,
                                         solver="exact", sample_normalize=False)
        assert abs(dens[:, 1].sum() - 532) < 1e-7 * 532, dens[:, 1].sum()
        out["g1_" + filt] = dens
    np.savez_compressed(os.path.join(HERE, "g1_exact_1000x2.npz"), data=data, labels=labels, samples=samples,
                        dens_heat=out["g1_heat"], dens_laplacian=out["g1_laplacian"])

    # G2: same data, default sparse Chebyshev path (knn=5, decay=40, thresh=1e-4, anisotropy=1, M=50)
    samples, dens, G = mo.fit_transform(data, labels, return_graph=True, algorithm="brute")
    h = mo.filter_kernel_fn("heat", 60, 0, 1, G.lmax)
    np.savez_compressed(os.path.join(HERE, "g2_cheby_1000x2.npz"), data=data, labels=labels, samples=samples, dens=dens,
                        coeffs=mo.cheby_coeff(h, G.lmax, 50), bandwidth=G.info["bandwidth"],
                        W_data=G.W.data, W_indices=G.W.indices.astype(np.int32), **graph_summary(G))

    # G3: README toy (README.md:51-57), exercises the radius-expansion path
    rng = np.random.default_rng(0)
    X = rng.normal(size=(500, 100))
    lab = rng.choice(["treatment", "control"], size=500)
    samples, dens, G = mo.fit_transform(X, lab, return_graph=True, algorithm="brute")
    np.savez_compressed(os.path.join(HERE, "g3_readme_500x100.npz"), x_sha=sha(X), labels=lab, samples=samples, dens=dens,
                        n_research_rows=np.int64(G.info["n_research_rows"]), **graph_summary(G))

    # G4: three labels (reference test/test_meld.py:169-172)
    rng = np.random.default_rng(4)
    X = rng.normal(size=(300, 2))
    lab = rng.choice(["A", "B", "C"], size=300)
    samples, dens, G = mo.fit_transform(X, lab, return_graph=True, algorithm="brute")
    np.savez_compressed(os.path.join(HERE, "g4_three_labels_300x2.npz"), data=X, labels=lab, samples=samples, dens=dens,
                        **graph_summary(G))

    # G5: make_batches(100) (reference test/utils/__init__.py:37-71), N = 600
    X, lab = make_batches(100, seed=0)
    samples, dens, G = mo.fit_transform(X, lab, return_graph=True, algorithm="brute")
    np.savez_compressed(os.path.join(HERE, "g5_batches_600x2.npz"), data=X, labels=lab, samples=samples, dens=dens,
                        **graph_summary(G))

    # G6: C2-shaped mini (N=5000, d=50, knn=15, M=30); inputs regenerated by mo.synthetic_cells(5000, 50, seed=0)
    X, lab = mo.synthetic_cells(5000, n_dims=50, seed=0)
    samples, dens, G = mo.fit_transform(X, lab, knn=15, beta=60, chebyshev_order=30, return_graph=True, algorithm="brute")
    np.savez_compressed(os.path.join(HERE, "g6_c2mini_5000x50.npz"), x_sha=sha(X), labels_sha=sha(lab.astype("U4")),
                        samples=samples, dens=dens, bandwidth=G.info["bandwidth"], **graph_summary(G))
    make_g7()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
