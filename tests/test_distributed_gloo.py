"""N > 1 path on CPU: world_size 2 and 3 over gloo.  The real communication code of
meld_amd.distributed (row sharding, all-to-all-v of the transposed edges, all-gather of the
kernel row sums / Lanczos vector / Chebyshev iterate, scalar all-reduces) runs on CPU tensors
with a NumPy stand-in for the per-GPU kernels (tests/cpu_ops.py); results must equal the
single-process oracle."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
from scipy import sparse

from oracle import meld_oracle as mo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, tmp_path, n, d, knn, n_labels, n_pca=0, extra_env=None):
    out = str(tmp_path / "res")
    from tests.conftest import run_ranks

    run_ranks("dist_worker.py", [out, n, d, knn, n_labels, n_pca], world, timeout=600, env=extra_env)
    return [np.load(out + ".rank{}.npz".format(r)) for r in range(world)]


# (world, n, labels, d, n_pca): 2 and 3 ranks; 8 ranks of which four own NO rows (shards are whole 256-row
# search workgroups: N = 1001 fills ranks 0..3 only) -- the empty ranks still have to take part in every
# collective; 4 ranks with a ragged tail (9 rows on the last rank); wide data through the PCA front end
# (n_pca < min(X.shape), the reference's default situation on gene-space input)
@pytest.mark.parametrize("world,n,n_labels,d,n_pca", [(2, 1001, 2, 8, 0), (3, 700, 3, 8, 0), (8, 1001, 2, 8, 0),
                                                      (4, 777, 2, 8, 0), (2, 600, 2, 40, 12)])
def test_sharded_fit_transform_equals_oracle(world, n, n_labels, d, n_pca, tmp_path):
    knn = 7
    ranks = _run(world, tmp_path, n, d, knn, n_labels, n_pca)
    X, labels = mo.synthetic_cells(n, n_dims=d, seed=7)
    if n_labels == 3:
        labels = np.random.default_rng(1).choice(["A", "B", "C"], size=n)
    G = mo.build_graph(X, knn=knn, algorithm="brute", n_pca=n_pca or None)
    # graph: the shards tile the oracle's W exactly
    W = sparse.vstack([
        sparse.csr_matrix((r["val"], r["col"], r["rowptr"][: int(r["n_rows"]) + 1]), shape=(int(r["n_rows"]), n)) for r in ranks
    ]).tocsr()
    per_rank = -(-(-(-n // world)) // 256) * 256  # shards are whole search workgroups (256 rows)
    assert [int(r["row_begin"]) for r in ranks] == [min(i * per_rank, n) for i in range(world)]
    assert [int(r["n_rows"]) for r in ranks] == [max(0, min(n, (i + 1) * per_rank) - min(n, i * per_rank)) for i in range(world)]
    assert W.nnz == G.W.nnz == int(ranks[0]["nnz_global"])
    # the transposed entries travelled in ONE fixed-capacity all-to-all (no split sizes read back on the host)
    assert all(str(r["exchange"]) == "fixed" and int(r["exchange_overflow"]) == 0 for r in ranks)
    wtol = 1e-13 if not n_pca else 1e-8  # PCA scores by two different routes (covariance eigh vs full SVD)
    assert abs(W - G.W).max() < wtol
    np.testing.assert_allclose(np.concatenate([r["dw"][: int(r["n_rows"])] for r in ranks]), G.dw, rtol=1e-12 if not n_pca else 1e-8)
    # lmax: every rank agrees, and it is the converged top eigenvalue x 1.01
    lam = float(sparse.linalg.eigsh(G.L, k=1, tol=1e-12, return_eigenvectors=False)[0])
    for r in ranks:
        assert float(r["lmax"]) == float(ranks[0]["lmax"])
    assert abs(float(ranks[0]["lmax"]) / 1.01 - lam) / lam < 5e-5  # default Lanczos tolerance 1e-3: eigenvalue error ~ tol^2
    # the estimate ran with ONE all-reduce per iteration and stops at the same prefix with the same value as the
    # two-all-reduce form of the iteration
    assert int(ranks[0]["lanczos_all_reduces"]) == 1
    assert int(ranks[0]["lanczos_iters"]) == int(ranks[0]["iters_unfolded"])
    assert abs(float(ranks[0]["lmax"]) / 1.01 - float(ranks[0]["theta_unfolded"])) < 1e-10 * lam
    # densities: identical on every rank and equal to the oracle with the same lmax
    samples, ind = mo.sample_indicators(labels)
    ref = mo.meld_filter(ind, G, beta=40, chebyshev_order=25, lmax=float(ranks[0]["lmax"]))
    for r in ranks:
        assert list(r["columns"]) == list(samples)
        assert np.abs(r["dens"] - ref).max() / np.abs(ref).max() < (1e-11 if not n_pca else 1e-6)
        np.testing.assert_array_equal(r["dens"], ranks[0]["dens"])


def test_exchange_overflow_falls_back_to_the_variable_length_exchange(tmp_path):
    """A capacity too small for what some rank owes a peer is seen by every rank (the overflow count rides in the build's
    last all-reduce) and the build repeats the exchange with split sizes: same graph, same densities."""
    n, d, knn, world = 700, 8, 7, 3
    ranks = _run(world, tmp_path, n, d, knn, 2, extra_env=dict(MELD_EXCHANGE_CAP="16"))
    assert all(str(r["exchange"]) == "variable" and int(r["exchange_overflow"]) > 0 for r in ranks)
    X, labels = mo.synthetic_cells(n, n_dims=d, seed=7)
    G = mo.build_graph(X, knn=knn, algorithm="brute")
    W = sparse.vstack([
        sparse.csr_matrix((r["val"], r["col"], r["rowptr"][: int(r["n_rows"]) + 1]), shape=(int(r["n_rows"]), n)) for r in ranks
    ]).tocsr()
    assert W.nnz == G.W.nnz and abs(W - G.W).max() < 1e-13
    samples, ind = mo.sample_indicators(labels)
    ref = mo.meld_filter(ind, G, beta=40, chebyshev_order=25, lmax=float(ranks[0]["lmax"]))
    for r in ranks:
        assert np.abs(r["dens"] - ref).max() / np.abs(ref).max() < 1e-11


@pytest.mark.parametrize("mode", ["unweighted", "replicated", "cosine"])
def test_sharded_breadth_unweighted_graph_and_replicated_build(mode, tmp_path):
    """(a) decay=None (graphtools' unweighted kNN graph, forwarded at reference meld/meld.py:106,118) on the row-sharded
    builder; (b) a graph every rank holds in full -- the route of the MNN graph (sample_idx) and of graphs built elsewhere --
    sharded for the recurrences only (shard_of_graph); (c) distance="cosine" (the metric enters through the data on every rank:
    graph.metric_front_end).  3 ranks, ragged tail; graph and densities against the oracle."""
    n, d, knn, world = 700, 8, 7, 3
    ranks = _run(world, tmp_path, n, d, knn, 2, extra_env=dict(MELD_TEST_MODE=mode))
    X, labels = mo.synthetic_cells(n, n_dims=d, seed=7)
    G = mo.build_graph(X, knn=knn, algorithm="brute", decay=None if mode == "unweighted" else 40,
                       distance="cosine" if mode == "cosine" else "euclidean")
    W = sparse.vstack([
        sparse.csr_matrix((r["val"], r["col"], r["rowptr"][: int(r["n_rows"]) + 1]), shape=(int(r["n_rows"]), n)) for r in ranks
    ]).tocsr()
    assert W.nnz == G.W.nnz == int(ranks[0]["nnz_global"]) and abs(W - G.W).max() < 1e-13
    samples, ind = mo.sample_indicators(labels)
    ref = mo.meld_filter(ind, G, beta=40, chebyshev_order=25, lmax=float(ranks[0]["lmax"]))
    for r in ranks:
        assert float(r["lmax"]) == float(ranks[0]["lmax"])
        assert np.abs(r["dens"] - ref).max() / np.abs(ref).max() < 1e-11
        np.testing.assert_array_equal(r["dens"], ranks[0]["dens"])


def test_shard_range_covers_everything():
    from meld_amd.distributed import shard_range

    for n, w in [(10, 3), (7, 8), (1000, 8), (64, 2)]:
        got = []
        for r in range(w):
            R, b, c = shard_range(n, w, r)
            got += list(range(b, b + c))
            assert c <= R
        assert got == list(range(n))


def test_sharded_filterbank_vfc(tmp_path):
    """BASELINE configs[4] (VertexFrequencyCluster on the sharded driver): the filter-bank method on a row-sharded graph
    -- iterate all-gathered per SpMM, Gram matrices of the CholeskyQR / Rayleigh-Ritz / deflation all-reduced -- gives
    every rank the same spectrogram, the same as one rank computes and as the code path without a process group.
    (transform / predict work on the gathered spectrogram and are the single-GPU code.)"""
    res = {}
    for world in (1, 2, 3):
        out = str(tmp_path / "vfc{}".format(world))
        from tests.conftest import run_ranks

        run_ranks("dist_worker_vfc.py", [out, 900, 6, 7], world)
        res[world] = [np.load(out + ".rank{}.npz".format(k)) for k in range(world)]
    ref = res[1][0]
    assert ref["spec"].shape == (900, 24 + 6) and np.isfinite(ref["spec"]).all() and ref["spec"].min() >= 0.0
    assert np.abs(ref["spec"] - ref["spec_unsharded"]).max() < 1e-9
    for world in (2, 3):
        for r in res[world]:
            assert abs(float(r["lmax"]) - float(ref["lmax"])) < 1e-9 * float(ref["lmax"])
            assert np.abs(r["spec"] - ref["spec"]).max() < 1e-7  # (lmax itself differs in the last digits between shardings)
            assert np.abs(r["ritz"] - ref["ritz"]).max() < 1e-7 * float(ref["lmax"])
            assert np.abs(r["norm2"] - ref["norm2"]).max() < 1e-7 * np.abs(ref["norm2"]).max()


def test_sharded_filterbank_vfc_wide_probe_block(tmp_path):
    """The same with 40 probes: more than 32 columns take the wide step (row-major iterate, ONE all-gather per product instead of
    one per column pair); world 1 and 2 agree, world 1 equals the path without a process group."""
    res = {}
    for world in (1, 2):
        out = str(tmp_path / "vfcw{}".format(world))
        from tests.conftest import run_ranks

        run_ranks("dist_worker_vfc.py", [out, 900, 6, 7], world, env=dict(MELD_TEST_PROBES="40"))
        res[world] = [np.load(out + ".rank{}.npz".format(k)) for k in range(world)]
    ref = res[1][0]
    assert ref["spec"].shape == (900, 40 + 6) and np.isfinite(ref["spec"]).all()
    assert np.abs(ref["spec"] - ref["spec_unsharded"]).max() < 1e-9
    for r in res[2]:
        assert np.abs(r["spec"] - ref["spec"]).max() < 1e-7
        assert np.abs(r["ritz"] - ref["ritz"]).max() < 1e-7 * float(ref["lmax"])
