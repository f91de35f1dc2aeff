"""GPU tests of the estimator surface, restating reference test/test_meld.py:108-192 and
test/test_utils.py:9-31 (API / error-message contract) against meld_amd, plus the committed
golden fixtures (tests/golden/*.npz; tolerance 1e-5 relative to the column maximum, the bound
BASELINE.json's north_star states -- measured ~1e-12)."""
import os

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _meld():
    import meld_amd

    return meld_amd


def _oracle():
    from oracle import meld_oracle

    return meld_oracle


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def _close(out, ref, tol=1e-5):
    out = np.asarray(out)
    for c in range(ref.shape[1]):
        assert np.abs(out[:, c] - ref[:, c]).max() / np.abs(ref[:, c]).max() < tol


def test_labels_wrong_shape():
    meld = _meld()
    data = np.random.normal(0, 2, (100, 2))
    sample_labels = np.ones([101, 2], dtype=str)
    with pytest.raises(ValueError) as e:
        meld.MELD(verbose=0).fit_transform(X=data, sample_labels=sample_labels)
    assert str(e.value) == "Input data ({}) and input graph ({}) " "are not of the same size".format(sample_labels.shape, 100)


def test_label_2d_column_dataframe():
    meld = _meld()
    data = np.random.normal(0, 2, (100, 2))
    index = pd.Index(["cell_{}".format(i) for i in range(100)])
    sample_labels = pd.DataFrame(np.concatenate([np.zeros((50, 1)), np.ones((50, 1))]), index=index, columns=pd.Index(["A"]), dtype=str)
    out = meld.MELD(verbose=0).fit_transform(X=data, sample_labels=sample_labels)
    assert out.shape == (100, 2)


def test_label_dataframe_index_and_columns_preserved():
    meld = _meld()
    data = np.random.normal(0, 2, (100, 2))
    index = pd.Index(["cell_{}".format(i) for i in range(100)])
    sample_labels = pd.DataFrame(np.concatenate([np.zeros(50), np.ones(50)]), index=index, columns=["sample_labels"], dtype=str)
    out = meld.MELD(verbose=0).fit_transform(X=data, sample_labels=sample_labels)
    assert np.all(out.index == index)
    assert np.all(out.columns == pd.Index(np.unique(sample_labels)))


def test_labels_non_numeric_two_and_three():
    meld = _meld()
    data = np.random.normal(size=(100, 2))
    out = meld.MELD().fit_transform(data, np.random.choice(["A", "B"], size=100))
    assert list(out.columns) == ["A", "B"]
    out = meld.MELD().fit_transform(data, np.random.choice(["A", "B", "C"], size=100))
    assert np.all(out.columns == ["A", "B", "C"])
    np.testing.assert_allclose(out.values.sum(0), 1.0, rtol=1e-4)


def test_one_sample_message():
    meld = _meld()
    data = np.random.normal(size=(100, 2))
    with pytest.raises(ValueError) as e:
        meld.MELD().fit_transform(data, np.ones(100))
    assert str(e.value) == "Found only one unqiue sample label. Cannot estimate density " "of a single sample."


def test_reset_semantics_with_a_real_graph():
    """reference test/test_meld.py:83-93"""
    meld = _meld()
    data = np.random.normal(0, 2, (500, 2))
    labels = np.random.choice(["ctrl", "treat"], size=500)
    op = meld.MELD(verbose=0, knn=20, decay=10)
    op.fit_transform(data, labels)
    g0 = op.graph
    op.set_params(beta=op.beta + 1)
    assert op.sample_densities is None and op.graph is g0
    op.fit_transform(data, labels)
    assert op.sample_densities is not None and op.graph is g0  # same data: the graph is reused
    op.set_params(knn=op.knn + 1)
    assert op.graph is None and op.sample_densities is None


def test_prebuilt_graph_and_repeated_transform():
    """fit(G) then transform(labels) many times (reference meld/benchmark.py:194-195)."""
    meld = _meld()
    from oracle import meld_oracle as mo

    X, labels = mo.synthetic_cells(2000, n_dims=20, seed=4)
    op = meld.MELD(knn=10, chebyshev_order=30)
    op.fit(X)
    G = op.graph
    op2 = meld.MELD(chebyshev_order=30).fit(G)
    a = op2.transform(labels)
    b = op.transform(labels)
    np.testing.assert_allclose(a.values, b.values, rtol=0, atol=0)
    G2 = meld.DeviceGraph.from_scipy(G.W)  # a graph built elsewhere, uploaded
    G2.lmax = G.lmax
    c = meld.MELD(chebyshev_order=30).fit(G2).transform(labels)
    np.testing.assert_allclose(c.values, a.values, rtol=1e-12)
    lik = meld.utils.normalize_densities(a)
    np.testing.assert_allclose(lik.values.sum(1), 1.0)


def test_fit_adopts_a_graph_built_elsewhere():
    """reference meld/utils.py:11-20, meld/benchmark.py:194-195, test/test_utils.py:11-13: ``fit(G)`` /
    ``fit_transform(G, labels)`` with a graphtools / pygsp graph object.  Here any object exposing a square
    scipy-sparse ``.W`` (and optionally a cached ``_lmax``) is uploaded; the oracle-built graph stands in for
    the foreign one.  The result equals the oracle's filter on that very graph."""
    meld = _meld()
    from oracle import meld_oracle as mo

    X, labels = mo.synthetic_cells(2500, n_dims=20, seed=6)
    G = mo.build_graph(X, knn=10, algorithm="brute")
    lmax = mo.estimate_lmax(G.L, G.dw)

    class ForeignGraph:  # what graphtools' PyGSP graph offers on this path
        def __init__(self, W, lm):
            self.W, self.N, self._lmax = W, W.shape[0], lm

    fg = ForeignGraph(G.W, lmax)
    op = meld.MELD(chebyshev_order=30)
    out = op.fit_transform(fg, labels)
    assert isinstance(op.graph, meld.DeviceGraph) and op.graph.N == 2500 and op.graph.lmax == lmax
    samples, ind = mo.sample_indicators(labels)
    ref = mo.meld_filter(ind, G, beta=60, chebyshev_order=30, lmax=lmax)
    assert list(out.columns) == list(samples)
    assert np.abs(out.values - ref).max() <= 1e-5 * np.abs(ref).max()
    # the guard itself: converts, and keeps raising the reference's TypeError for non-graphs
    assert isinstance(meld.utils._check_pygsp_graph(fg), meld.DeviceGraph)
    for bad in ("hello world", np.zeros((3, 3)), G.W):
        with pytest.raises(TypeError, match="Input graph should be of type graphtools.base.BaseGraph"):
            meld.utils._check_pygsp_graph(bad)
    # VertexFrequencyCluster on an uploaded graph without its kernel (ksum unknown: diagonal 1)
    sub = ForeignGraph(G.W[:600][:, :600].tocsr(), None)
    vfc = meld.VertexFrequencyCluster(n_clusters=2, window_sizes=[1, 2]).fit(sub)
    assert vfc.N == 600
    # raw data after an adopted graph rebuilds instead of keeping the stale graph
    X2, _ = mo.synthetic_cells(2500, n_dims=20, seed=7)
    op.fit(X2)
    assert op.graph.info.get("adopted_from") is None and op.graph.N == 2500
    assert abs(op.graph.W - mo.build_graph(X2, knn=5, algorithm="brute").W).max() < 1e-9


def test_arpack_lmax_mode_reproduces_the_reference_estimate():
    """``MELD(lmax="arpack")``: pygsp's estimate itself (eigsh(L, k=1, tol=5e-3, ncv=10) * 1.01, reference
    meld/filter.py:39) evaluated on the host, so that an UN-injected fit_transform lands on the reference's
    numbers.  ARPACK's start vector comes from process-global state, so both sides are evaluated in fresh
    processes (first eigsh call of the process each): they then agree to rounding and the densities meet the
    1e-5 bar without a common injected lmax."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code_ref = (
        "import numpy as np, sys; sys.path.insert(0, %r)\n"
        "from oracle import meld_oracle as mo\n"
        "X, labels = mo.synthetic_cells(3000, n_dims=50, seed=3)\n"
        "s, d, G = mo.fit_transform(X, labels, knn=15, chebyshev_order=30, return_graph=True, algorithm='brute')\n"
        "np.save(sys.argv[1], d); print(repr(G.lmax))\n" % root)
    code_dev = (
        "import numpy as np, sys; sys.path.insert(0, %r)\n"
        "import meld_amd\nfrom bench import synthetic_cells\n"
        "X, labels = synthetic_cells(3000, 50, seed=3)\n"
        "op = meld_amd.MELD(knn=15, chebyshev_order=30, lmax='arpack')\n"
        "d = op.fit_transform(X, labels)\n"
        "np.save(sys.argv[1], d.values); print(repr(op.graph.lmax)); print(op.graph.lmax_info['method'])\n" % root)
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        a = subprocess.run([sys.executable, "-c", code_ref, os.path.join(tmp, "ref.npy")], capture_output=True, text=True, timeout=600)
        b = subprocess.run([sys.executable, "-c", code_dev, os.path.join(tmp, "dev.npy")], capture_output=True, text=True, timeout=600)
        assert a.returncode == 0, a.stderr[-2000:]
        assert b.returncode == 0, b.stderr[-2000:]
        lm_ref = float(a.stdout.strip().splitlines()[-1])
        lines = [ln for ln in b.stdout.strip().splitlines() if ln.strip()]
        lm_dev = float(lines[-2])
        assert lines[-1] == "arpack"
        ref, dev = np.load(os.path.join(tmp, "ref.npy")), np.load(os.path.join(tmp, "dev.npy"))
    assert abs(lm_dev - lm_ref) <= 1e-9 * lm_ref, (lm_dev, lm_ref)
    assert np.abs(dev - ref).max() <= 1e-5 * np.abs(ref).max()
    with pytest.raises(ValueError, match="not recognized"):
        _meld().MELD(lmax="power").fit_transform(np.random.default_rng(0).normal(size=(200, 3)), np.arange(200) % 2)


def test_unsupported_options_fail_loudly():
    meld = _meld()
    data = np.random.normal(size=(100, 2))
    labels = np.random.choice(["a", "b"], size=100)
    out = meld.MELD(verbose=0).fit_transform(data, labels, sample_idx=labels)  # MNN graph (reference test_mnn): supported
    assert out.shape == (100, 2) and np.isfinite(out.values).all()
    # n_landmark (reference meld/meld.py:105): a LandmarkGraph's kernel / Laplacian are those of the plain graph, so the
    # densities are the same; only the landmark operator itself is not offered
    a = meld.MELD(n_landmark=50, verbose=0).fit_transform(data, labels)
    b = meld.MELD(verbose=0).fit_transform(data, labels)
    assert np.array_equal(a.values, b.values)
    with pytest.raises(NotImplementedError):
        meld.MELD(n_landmark=50, verbose=0).fit(data).graph.landmark_op
    with pytest.raises(NotImplementedError):
        meld.MELD(verbose=0).fit(data, search_multiplier=3)  # a graph keyword the builder does not know (bandwidth, knn_max, kernel_symm ... it does)
    with pytest.raises(NotImplementedError):
        meld.MELD(verbose=0).fit(data, kernel_symm=None)  # a directed kernel
    with pytest.raises(ValueError):
        meld.MELD(distance="mahalanobis")  # (euclidean-reducible metrics on the search kernel, manhattan / chebyshev densely)


@pytest.mark.gpu
@pytest.mark.parametrize("metric", ["manhattan", "chebyshev"])
def test_metrics_outside_the_euclidean_search_take_the_dense_route(metric):
    """manhattan / chebyshev: no function of the euclidean distance, so the matrix-pipe search does not apply; the same kernel on
    library pairwise distances (small N), against the oracle, which hands the metric to sklearn."""
    meld = _meld()
    mo = _oracle()
    X, labels = mo.synthetic_cells(1800, n_dims=6, seed=21)
    G = mo.build_graph(X, knn=7, algorithm="brute", distance=metric)
    op = meld.MELD(knn=7, distance=metric, verbose=0).fit(X)
    W = op.graph.W
    assert W.nnz == G.W.nnz and abs(W - G.W).max() <= 1e-9 * abs(G.W).max()
    big = np.zeros((20000, 3))
    with pytest.raises(NotImplementedError):
        meld.MELD(distance=metric, verbose=0).fit(big)


@pytest.mark.gpu
@pytest.mark.parametrize("metric", ["correlation", "sqeuclidean", "l2"])
def test_other_reducible_metrics_match_the_oracle(metric):
    """correlation = cosine of the rows with their means removed, sqeuclidean = euclidean with the decay doubled, l2 = euclidean:
    each against the oracle, which hands the metric to sklearn."""
    from scipy import sparse

    meld = _meld()
    mo = _oracle()
    rng = np.random.default_rng(17)
    n = 4000
    X = rng.normal(size=(n, 10)) * rng.uniform(0.5, 3.0, size=(n, 1)) + rng.normal(size=(n, 1))
    op = meld.MELD(knn=6, decay=20, distance=metric, n_pca=None, verbose=0).fit(X)
    G = mo.build_graph(X, knn=6, decay=20, distance=metric)
    A, B = sparse.csr_matrix(op.graph.W), sparse.csr_matrix(G.W)
    A.sort_indices(); B.sort_indices()
    assert A.nnz == B.nnz and np.array_equal(A.indices, B.indices)
    np.testing.assert_allclose(A.data, B.data, rtol=1e-9)
    np.testing.assert_allclose(op.graph.bandwidth_host, G.info["bandwidth"], rtol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("decay,thresh,n", [(40, 1e-4, 6000), (None, 1e-4, 3000), (10, 0, 700)])
def test_cosine_distance_matches_the_oracle(decay, thresh, n):
    """distance="cosine" ([UPSTREAM graphtools kNNGraph(distance=...)] -> sklearn's cosine metric): the oracle hands the metric to
    sklearn, the product builds the euclidean graph of the unit rows with the decay doubled (graph.metric_front_end).  Same
    pattern, weights and densities; bandwidths are reported in cosine units; all-zero rows are refused."""
    from scipy import sparse

    meld = _meld()
    mo = _oracle()
    rng = np.random.default_rng(n)
    X = rng.normal(size=(n, 12)) * rng.uniform(0.2, 5.0, size=(n, 1)) + 1.5  # rows of very different lengths
    labels = np.where(rng.random(n) < 0.4, "a", "b")
    op = meld.MELD(knn=7, decay=decay, thresh=thresh, distance="cosine", n_pca=None, verbose=0)
    dens = op.fit_transform(X, labels)
    if thresh == 0:
        from scipy.spatial.distance import pdist, squareform

        D = squareform(pdist(X, metric="cosine"))
        bw = np.sort(D, axis=1)[:, 7]
        Kd = np.exp(-((D / bw[:, None]) ** decay))
        K = 0.5 * (Kd + Kd.T)
        ks = K.sum(axis=1)
        W = K / np.outer(ks, ks)
        np.fill_diagonal(W, 0.0)
        got = op.graph.W.toarray()
        np.testing.assert_allclose(got, W, rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(op.graph.bandwidth_host, bw, rtol=1e-10)
        return
    G = mo.build_graph(X, knn=7, decay=decay, thresh=thresh, distance="cosine")
    A, B = sparse.csr_matrix(op.graph.W), sparse.csr_matrix(G.W)
    A.sort_indices(); B.sort_indices()
    assert A.nnz == B.nnz and np.array_equal(A.indices, B.indices)
    np.testing.assert_allclose(A.data, B.data, rtol=1e-9)
    np.testing.assert_allclose(op.graph.bandwidth_host, G.info["bandwidth"], rtol=1e-10)
    samples, ref = mo.fit_transform(X, labels, knn=7, decay=decay, thresh=thresh, distance="cosine", lmax=op.graph.lmax)
    np.testing.assert_allclose(dens.values, ref, rtol=1e-7)
    Xz = X.copy()
    Xz[5] = 0.0
    with pytest.raises(ValueError):
        meld.MELD(distance="cosine", n_pca=None, verbose=0).fit(Xz)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,n_pca", [((3000, 60), 20), ((600, 900), 15), ((2500, 1200), 10)])
def test_pca_front_end_matches_the_oracle(shape, n_pca):
    """n_pca < min(X.shape): the graph is built on the PCA scores (graphtools Data._reduce_data).
    The product computes the exact subspace (covariance / Gram eigh, randomized range finder above
    pca.EXACT_MAX); the oracle sklearn's full-SVD PCA.  Distances -- all the graph sees -- are
    invariant to the sign conventions, so W and the densities must agree to rounding."""
    import torch

    from meld_amd import pca as mpca

    meld = _meld()
    mo = _oracle()
    rng = np.random.default_rng(3)
    N, G = shape
    latent = rng.normal(size=(N, 8)) * np.array([9, 7, 5, 4, 3, 2.5, 2, 1.5])
    X = latent @ rng.normal(size=(8, G)) + 0.05 * rng.normal(size=(N, G)) + rng.normal(size=G)
    labels = np.where(latent[:, 0] + rng.normal(size=N) > 0, "treat", "ctrl")
    old = mpca.EXACT_MAX
    try:
        if shape == (2500, 1200):
            mpca.EXACT_MAX = 1000  # force the randomized range finder
        Y = mpca.pca_project(torch.from_numpy(X).cuda(), n_pca).cpu().numpy()
        Yr = mo.pca_reduce(X, n_pca)
        from scipy.spatial.distance import pdist

        sub = rng.choice(N, size=400, replace=False)
        tol = 1e-9 if shape != (2500, 1200) else 1e-6  # the sketch converges geometrically, not exactly
        assert np.abs(pdist(Y[sub]) - pdist(Yr[sub])).max() <= tol * pdist(Yr[sub]).max()
        if shape == (2500, 1200):
            return
        op = meld.MELD(n_pca=n_pca, knn=7, chebyshev_order=30, verbose=0)
        G_ref = mo.build_graph(X, knn=7, n_pca=n_pca)
        op.fit(X)
        W = op.graph.W
        assert (W != 0).multiply(G_ref.W != 0).nnz == W.nnz == G_ref.W.nnz
        assert abs(W - G_ref.W).max() <= 1e-8
        lmax = mo.estimate_lmax(G_ref.L, G_ref.dw)
        op.graph.lmax = lmax
        dens = op.transform(labels)
        ref = mo.meld_filter(mo.sample_indicators(labels)[1], G_ref, beta=60, chebyshev_order=30, lmax=lmax)
        assert np.abs(dens.values - ref).max() <= 1e-5 * np.abs(ref).max()
    finally:
        mpca.EXACT_MAX = old


@pytest.mark.parametrize(
    "fixture,kw",
    [
        ("g2_cheby_1000x2.npz", {}),
        ("g4_three_labels_300x2.npz", {}),
        ("g5_batches_600x2.npz", {}),
    ],
)
def test_golden_small(fixture, kw):
    meld = _meld()
    g = load(fixture)
    op = meld.MELD(lmax=float(g["lmax"]), **kw)
    out = op.fit_transform(g["data"], g["labels"])
    assert list(out.columns) == list(g["samples"])
    assert op.graph.nnz == int(g["nnz"])
    np.testing.assert_allclose(op.graph.dw, g["dw"], rtol=1e-9)
    assert np.array_equal(op.graph.W.indptr, g["rowptr"])
    _close(out.values, g["dens"])
    if "W_data" in g.files:
        np.testing.assert_allclose(op.graph.W.data, g["W_data"], rtol=1e-9)
        assert np.array_equal(op.graph.W.indices, g["W_indices"])
        np.testing.assert_allclose(op.graph.bandwidth_host, g["bandwidth"], rtol=1e-12)


@pytest.mark.gpu
def test_golden_graph_options():
    """G7: graphtools' bandwidth_scale / bandwidth / knn_max / kernel_symm through the estimator against the committed
    fixture (oracle-derived; tools/regen_golden_from_reference.py hands the same keywords to the real stack)."""
    meld = _meld()
    from tests.golden import make_golden as mg

    g = load("g7_graph_options_1000x8.npz")
    X, labels = mg.g7_inputs()
    assert mg.sha(X) == str(g["x_sha"])
    for tag, kw in mg.G7_OPTIONS:
        op = meld.MELD(knn=7, chebyshev_order=30, lmax=float(g[tag + "_lmax"]), verbose=0, **kw)
        out = op.fit_transform(X, labels)
        W = op.graph.W
        assert W.nnz == int(g[tag + "_nnz"]) and np.array_equal(W.indptr, g[tag + "_rowptr"]) and np.array_equal(W.indices, g[tag + "_W_indices"]), tag
        np.testing.assert_allclose(W.data, g[tag + "_W_data"], rtol=1e-9)
        np.testing.assert_allclose(op.graph.dw, g[tag + "_dw"], rtol=1e-9)
        _close(out.values, g[tag + "_dens"])


def test_golden_readme_toy():
    meld = _meld()
    from tests.golden.make_golden import sha

    g = load("g3_readme_500x100.npz")
    rng = np.random.default_rng(0)
    X = rng.normal(size=(500, 100))
    lab = rng.choice(["treatment", "control"], size=500)
    assert sha(X) == str(g["x_sha"]) and np.array_equal(lab, g["labels"])
    op = meld.MELD(lmax=float(g["lmax"]))
    out = op.fit_transform(X, lab)
    assert op.graph.nnz == int(g["nnz"])
    np.testing.assert_allclose(op.graph.dw, g["dw"], rtol=1e-9)
    _close(out.values, g["dens"])


def test_golden_c2_mini():
    meld = _meld()
    from oracle import meld_oracle as mo
    from tests.golden.make_golden import sha

    g = load("g6_c2mini_5000x50.npz")
    X, labels = mo.synthetic_cells(5000, n_dims=50, seed=0)
    assert sha(X) == str(g["x_sha"])
    op = meld.MELD(knn=15, beta=60, chebyshev_order=30, lmax=float(g["lmax"]))
    out = op.fit_transform(X, labels)
    assert op.graph.nnz == int(g["nnz"])
    np.testing.assert_allclose(op.graph.dw, g["dw"], rtol=1e-9)
    np.testing.assert_allclose(op.graph.bandwidth_host, g["bandwidth"], rtol=1e-12)
    _close(out.values, g["dens"])
    # native lmax lies within the reference's own run-to-run band of the ARPACK value
    op2 = meld.MELD(knn=15, beta=60, chebyshev_order=30)
    out2 = op2.fit_transform(X, labels)
    assert abs(op2.graph.lmax - float(g["lmax"])) / float(g["lmax"]) < 5e-3
    _close(out2.values, g["dens"], tol=2e-3)


@pytest.mark.parametrize("filt", ["heat", "laplacian"])
def test_reference_known_answer_532_through_the_product(filt):
    """The reference's only known-answer test on this path (test/test_meld.py:43-81), replayed
    through meld_amd on the GPU: thresh=0 dense graph + solver='exact' -> sum(density['treat'])
    == 532; plus pointwise agreement with the oracle fixture G1."""
    meld = _meld()
    from tests.golden.make_golden import g1_inputs

    data, sample_labels = g1_inputs()
    op = meld.MELD(verbose=0, knn=20, decay=10, thresh=0, anisotropy=0, filter=filt, solver="exact", sample_normalize=False)
    densities = op.fit_transform(data, sample_labels)
    expt_density = densities.iloc[:, 1]
    assert list(densities.columns) == ["ctrl", "treat"]
    np.testing.assert_allclose(np.sum(expt_density), 532)
    g = load("g1_exact_1000x2.npz")
    _close(densities.values, g["dens_" + filt], tol=1e-8)
    # reset semantics of the same reference test (:83-93)
    op.set_params(beta=op.beta + 1)
    assert op.sample_densities is None
    op.fit_transform(data, sample_labels)
    assert op.sample_densities is not None
    op.set_params(knn=op.knn + 1)
    assert op.graph is None and op.sample_densities is None


def test_exact_solver_on_sparse_graph_matches_chebyshev():
    meld = _meld()
    from oracle import meld_oracle as mo

    X, labels = mo.synthetic_cells(1500, n_dims=10, seed=8)
    op = meld.MELD(knn=8, beta=20, solver="exact")
    a = op.fit_transform(X, labels)
    lmax_exact = op.graph.lmax  # the exact solver replaces lmax by the true top eigenvalue, like pygsp
    lam = float(__import__("scipy.sparse.linalg", fromlist=["eigsh"]).eigsh(op.graph.L, k=1, tol=1e-12, return_eigenvectors=False)[0])
    assert abs(lmax_exact - lam) / lam < 1e-10
    b = meld.MELD(knn=8, beta=20, solver="chebyshev", chebyshev_order=120, lmax=lmax_exact).fit_transform(X, labels)
    assert np.abs(a.values - b.values).max() / np.abs(a.values).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["U5", "U12", "S8", "int64"])
def test_device_label_factorization_matches_the_host_path(kind):
    """Large label arrays are factorised on the GPU (meld.MELD._factorize_device); codes, sorted
    uniques and counts must equal the host path's (= np.unique(..., return_inverse=True))."""
    import torch

    import meld_amd

    rng = np.random.default_rng(5)
    n = 300_000
    if kind == "int64":
        pool = np.array([7, -3, 2**40, 0, 11], dtype=np.int64)
    elif kind == "S8":
        pool = np.array([b"ctrl", b"treat", b"treat2", b"a"], dtype="S8")
    else:
        pool = np.array(["ctrl", "treat", "trea", "b", "zz"], dtype="<" + kind)
    labels = pool[rng.integers(0, pool.shape[0], size=n)]
    codes_h, uniq_h = meld_amd.MELD._factorize(labels)
    out = meld_amd.MELD._factorize_device(labels, torch.device("cuda"))
    assert out is not None
    codes_d, uniq_d, counts = out
    ref_u, ref_inv = np.unique(labels, return_inverse=True)
    assert np.array_equal(uniq_d, ref_u) and np.array_equal(uniq_h, ref_u)
    assert np.array_equal(codes_d.cpu().numpy(), ref_inv) and np.array_equal(codes_h, ref_inv)
    assert np.array_equal(counts, np.bincount(ref_inv))


@pytest.mark.gpu
def test_device_label_dictionary_limits_and_edge_cases():
    """meld_factorize_labels: up to 64 distinct labels of up to 16 words; beyond either the device path declines
    (None -> the host factorises); a label met only in the last, ragged chunk; labels that differ in their last word only."""
    import torch

    import meld_amd

    dev = torch.device("cuda")
    rng = np.random.default_rng(11)
    n = 250_123
    # 64 labels that share a long prefix and differ in the last characters (16 words = <U16)
    pool = np.array(["samplelabel_%04d" % i for i in range(64)], dtype="<U16")
    labels = pool[rng.integers(0, 64, size=n)]
    labels[-1] = pool[63]
    labels[:-1][labels[:-1] == pool[63]] = pool[0]  # label 63 occurs exactly once, in the ragged tail
    codes_d, uniq_d, counts = meld_amd.MELD._factorize_device(labels, dev)
    ref_u, ref_inv = np.unique(labels, return_inverse=True)
    assert np.array_equal(uniq_d, ref_u) and np.array_equal(codes_d.cpu().numpy(), ref_inv)
    assert np.array_equal(counts, np.bincount(ref_inv)) and counts[list(ref_u).index(pool[63])] == 1
    # 65 distinct labels: more than the dictionary holds
    pool65 = np.array(["l%03d" % i for i in range(65)], dtype="<U4")
    assert meld_amd.MELD._factorize_device(pool65[rng.integers(0, 65, size=n)], dev) is None
    # labels wider than 16 words
    wide = np.array(["a" * 20, "b" * 20], dtype="<U20")
    assert meld_amd.MELD._factorize_device(wide[rng.integers(0, 2, size=n)], dev) is None
    # one single label (transform's own check reports it): one group
    codes_d, uniq_d, counts = meld_amd.MELD._factorize_device(np.full(n, "only", dtype="<U4"), dev)
    assert list(uniq_d) == ["only"] and int(counts[0]) == n and not codes_d.any()


@pytest.mark.gpu
@pytest.mark.parametrize("p", [1, 2, 3, 7])
def test_indicator_signal_and_row_scatter_kernels(p):
    """meld_indicator_signal = zeros / one-hot scatter / index_select(perm) / zero padding in one pass;
    meld_scatter_rows_f64 = out[perm] = in."""
    import torch

    from meld_amd import filter as mfilter
    from meld_amd._lib import check, get_lib, ptr

    rng = np.random.default_rng(p)
    n, n_pad = 70_001, 70_144
    codes = rng.integers(0, p, size=n)
    scale = rng.random(p) + 0.5
    perm = torch.from_numpy(rng.permutation(n)).cuda()
    for sc in (None, scale):
        sig = mfilter.IndicatorSignal(codes, p, sc)
        ref = torch.from_numpy(sig.to_dense()).cuda()[perm]
        out = sig.to_device_ordered(torch.device("cuda"), perm, n_pad)
        assert out.shape == (n_pad, p) and torch.equal(out[:n], ref) and not out[n:].any()
        out = mfilter.IndicatorSignal(torch.from_numpy(codes).cuda(), p, sc).to_device_ordered(torch.device("cuda"), None, n)
        assert torch.equal(out, torch.from_numpy(sig.to_dense()).cuda())
    r = torch.from_numpy(rng.normal(size=(n, p))).cuda()
    back = torch.empty_like(r)
    check(get_lib().meld_scatter_rows_f64(ptr(r), ptr(perm), n, p, ptr(back), torch.cuda.current_stream().cuda_stream), "scatter")
    ref = torch.empty_like(r)
    ref[perm] = r
    assert torch.equal(back, ref)


@pytest.mark.gpu
def test_float32_input_is_widened_on_the_device():
    """float32 data (PCA scores usually are) crosses PCIe as float32 and is widened on the device: the same graph and
    densities, bit for bit, as the float64 copy of the same values."""
    meld = _meld()
    rng = np.random.default_rng(3)
    X32 = rng.normal(size=(4000, 20)).astype(np.float32)
    labels = rng.choice(["a", "b"], size=4000)
    a = meld.MELD(knn=7, verbose=0)
    da = a.fit_transform(X32, labels)
    assert a.X.dtype == np.float32
    b = meld.MELD(knn=7, verbose=0, lmax=a.graph.lmax)
    db = b.fit_transform(X32.astype(np.float64), labels)
    import torch

    assert torch.equal(a.graph.val, b.graph.val) and torch.equal(a.graph.col, b.graph.col)
    np.testing.assert_allclose(da.values, db.values, rtol=1e-12, atol=1e-300)


@pytest.mark.gpu
def test_non_finite_input_is_rejected():
    meld = _meld()
    import torch

    X = np.random.default_rng(0).normal(size=(300, 5))
    X[17, 2] = np.nan
    with pytest.raises(ValueError, match="NaN or infinity"):
        meld.MELD(verbose=0).fit(X)
    Xt = torch.from_numpy(np.where(np.isnan(X), np.inf, X)).cuda()
    with pytest.raises(ValueError, match="NaN or infinity"):
        meld.MELD(verbose=0).fit(Xt)


@pytest.mark.gpu
def test_beta_sweep_in_one_pass_equals_separate_transforms():
    """Parameter-sweep mode: one recurrence pass, B accumulators."""
    meld = _meld()
    mo = _oracle()
    X, labels = mo.synthetic_cells(6000, n_dims=20, seed=13)
    labels = np.random.default_rng(2).choice(["a", "b", "c"], size=6000)
    op = meld.MELD(knn=7, chebyshev_order=30, verbose=0).fit(X)
    betas = [5, 20, 60, 150.5]
    sweep = op.transform_sweep(labels, betas)
    assert list(sweep) == betas
    for beta in betas:
        one = meld.MELD(knn=7, beta=beta, chebyshev_order=30, verbose=0).fit(op.graph).transform(labels)
        assert list(sweep[beta].columns) == list(one.columns)
        np.testing.assert_allclose(sweep[beta].values, one.values, rtol=0, atol=1e-13 * np.abs(one.values).max())
    from meld_amd import filter as mfilter

    s = np.random.default_rng(3).random((6000, 3))
    R = mfilter.filter_sweep(s, op.graph, "laplacian", [1.0, 7.0], order=2, chebyshev_order=25)
    for b, beta in enumerate([1.0, 7.0]):
        np.testing.assert_allclose(R[b], mfilter.filter(s, op.graph, "laplacian", beta, order=2, chebyshev_order=25), rtol=0, atol=1e-13)


def _two_batches(n_per=400, d=6, seed=0, shift=0.4):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n_per, d))
    b = rng.normal(size=(n_per + 57, d)) + shift
    X = np.concatenate([a, b])
    batch = np.array(["batch_a"] * n_per + ["batch_b"] * (n_per + 57))
    order = rng.permutation(X.shape[0])  # samples interleaved, as in real data
    return X[order], batch[order]


@pytest.mark.gpu
@pytest.mark.parametrize("decay", [40, None])
def test_mnn_graph_matches_the_oracle(decay):
    """sample_idx (reference test/test_meld.py:34, test/test_utils.py:11): graphtools' MNN kernel between
    samples -- same graph (pattern, weights 1e-9, degrees) and densities (1e-5 rel) as the oracle."""
    import meld_amd
    from oracle import meld_oracle as mo

    X, batch = _two_batches()
    rng = np.random.default_rng(5)
    labels = rng.choice(["ctrl", "expt"], size=X.shape[0])
    op = meld_amd.MELD(knn=7, decay=decay, chebyshev_order=30, verbose=0)
    dens = op.fit_transform(X, labels, sample_idx=batch)
    assert op.graph.info["graph"] == "mnn" and op.graph.info["n_samples"] == 2
    G = mo.build_graph(X, knn=7, decay=decay, sample_idx=batch, algorithm="brute")
    W = op.graph.W
    assert W.nnz == G.W.nnz and abs(W - G.W).max() <= 1e-9 * abs(G.W).max()
    np.testing.assert_allclose(op.graph.dw, G.dw, rtol=1e-9)
    lmax = mo.estimate_lmax(G.L, G.dw)
    op.graph.lmax = lmax
    dens = op.transform(labels)
    ref = mo.meld_filter(mo.sample_indicators(labels)[1], G, beta=60, chebyshev_order=30, lmax=lmax)
    assert np.abs(dens.values - ref).max() <= 1e-5 * np.abs(ref).max()


@pytest.mark.gpu
def test_mnn_graph_with_exact_subgraphs_matches_the_oracle():
    """sample_idx with thresh=0: graphtools builds the MNN kernel over dense "exact" subgraphs; same weights (1e-9) and
    densities (1e-5 rel) as the oracle's restatement."""
    import meld_amd
    from oracle import meld_oracle as mo

    X, batch = _two_batches()
    labels = np.random.default_rng(6).choice(["ctrl", "expt"], size=X.shape[0])
    op = meld_amd.MELD(knn=6, thresh=0, chebyshev_order=30, verbose=0)
    dens = op.fit_transform(X, labels, sample_idx=batch)
    assert op.graph.info["graph"] == "mnn" and op.graph.info["dense"]
    G = mo.build_graph(X, knn=6, thresh=0, sample_idx=batch)
    W = op.graph.W.toarray()
    assert abs(W - G.W).max() <= 1e-9 * abs(G.W).max()
    np.testing.assert_allclose(op.graph.dw, G.dw, rtol=1e-9)
    lmax = mo.estimate_lmax(G.L, G.dw)
    op.graph.lmax = lmax
    dens = op.transform(labels)
    ref = mo.meld_filter(mo.sample_indicators(labels)[1], G, beta=60, chebyshev_order=30, lmax=lmax)
    assert np.abs(dens.values - ref).max() <= 1e-5 * np.abs(ref).max()
    with pytest.raises(ValueError, match="more than one unique"):
        meld_amd.MELD(thresh=0, verbose=0).fit(X, sample_idx=np.zeros(X.shape[0]))


@pytest.mark.gpu
@pytest.mark.parametrize("nq,nr,d,knn,decay", [(700, 333, 6, 5, 40), (30000, 20011, 50, 15, 40), (4097, 9000, 20, 10, float("inf"))])
def test_cross_blocks_on_the_search_kernel_equal_the_library_path(nq, nr, d, knn, decay):
    """The blocks between two samples of the MNN kernel come from the MFMA search with the references restricted to
    the other sample (HipOps.directed_kernel_coo(n_refs=)): same entries as the fp64 GEMM + topk library path
    (cross_kernel, which test_oracle pins against the definition), values to 1e-12."""
    import torch
    from meld_amd.graph import HipOps
    from meld_amd.mnn import _cross_block, cross_kernel

    rng = np.random.default_rng(nq)
    lat = min(d, 8)
    A = rng.normal(size=(lat, d))
    Xq = torch.from_numpy(rng.normal(size=(nq, lat)) @ A + 0.05 * rng.normal(size=(nq, d))).cuda()
    Yr = torch.from_numpy((rng.normal(size=(nr, lat)) + 0.3) @ A + 0.05 * rng.normal(size=(nr, d))).cuda()  # a shifted batch
    ops = HipOps()
    r1, c1, v1 = _cross_block(ops, Xq, Yr, knn, decay, 1e-4)
    r0, c0, v0 = cross_kernel(Xq, Yr, knn, decay, 1e-4)
    k1 = torch.sort(r1 * nr + c1)
    k0 = torch.sort(r0 * nr + c0)
    assert k1.values.shape == k0.values.shape and torch.equal(k1.values, k0.values)
    assert float((v1[k1.indices] - v0[k0.indices]).abs().max()) <= 1e-12
    assert int(torch.bincount(r1, minlength=nq).min()) >= knn  # every query reaches its knn nearest references


@pytest.mark.gpu
def test_mnn_three_samples_and_argument_checks():
    import meld_amd
    from oracle import meld_oracle as mo

    rng = np.random.default_rng(11)
    X = np.concatenate([rng.normal(size=(150, 3)) + s for s in (0.0, 0.5, 1.0)])
    batch = np.repeat([0, 1, 2], 150)
    op = meld_amd.MELD(knn=4, verbose=0).fit(X, sample_idx=batch)
    G = mo.build_graph(X, knn=4, sample_idx=batch, algorithm="brute")
    W = op.graph.W
    assert W.nnz == G.W.nnz and abs(W - G.W).max() <= 1e-9 * abs(G.W).max()
    # the rest of the reference's test_mnn: likelihoods and vertex-frequency clustering on the MNN graph
    labels = np.where(rng.random(450) < 0.5, "ctrl", "expt")
    dens = op.transform(labels)
    lik = meld_amd.utils.normalize_densities(dens)
    spec = meld_amd.VertexFrequencyCluster(n_clusters=3, random_state=0).fit_transform(
        G=op.graph, sample_indicator=op.sample_indicators["expt"], likelihood=lik["expt"])
    ref_spec, _ = mo.vfc_transform(G.K, G.L, op.sample_indicators["expt"].values, likelihood=lik["expt"].values)
    assert spec.shape == ref_spec.shape and np.abs(spec - ref_spec).max() <= 1e-6
    with pytest.raises(ValueError, match="more than one unique value"):
        meld_amd.MELD(verbose=0).fit(X, sample_idx=np.zeros(450))
    with pytest.raises(ValueError, match="same length"):
        meld_amd.MELD(verbose=0).fit(X, sample_idx=batch[:-1])


def test_verbose_prints_the_stage_lines(capsys):
    """with verbose the build reports its stages the way graphtools does through tasklogger
    (reference notebooks/MELD_Quickstart.ipynb:190-198)."""
    meld = _meld()
    X = np.random.default_rng(0).normal(size=(3000, 10))
    meld.MELD(verbose=1, knn=7).fit(X)
    out = capsys.readouterr().out
    for piece in ("Building graph on 3000 samples and 10 features.", "Calculating graph and diffusion operator...",
                  "Calculating KNN search...", "Calculated KNN search in ", "Calculated affinities in ",
                  "Calculated graph and diffusion operator in "):
        assert piece in out, out
    meld.MELD(verbose=0, knn=7).fit(X)
    assert capsys.readouterr().out == ""


def test_graph_offers_what_the_reference_side_paths_read():
    """beyond N / lmax the graph object is asked for ``knn``, ``diff_op`` (graphtools; reference
    meld/cluster.py:213, comparison/comparison.py:318-323) and the Fourier basis ``U`` / ``e`` (pygsp; meld/cluster.py:235-236)
    -- checked against the oracle's graph (weights and kernel are the ones test_graph_matches_oracle pins)."""
    meld, mo = _meld(), _oracle()
    rng = np.random.default_rng(5)
    X = rng.normal(size=(900, 6)) * np.array([3.0, 2.0, 1.5, 1.0, 0.7, 0.5])
    op = meld.MELD(knn=7, verbose=0).fit(X)
    G = op.graph
    Go = mo.build_graph(X, knn=7)
    assert G.knn == 7
    P = G.diff_op
    Pref = Go.K.multiply(1.0 / np.ravel(Go.K.sum(1))[:, None]).tocsr()
    assert np.allclose(np.ravel(P.sum(1)), 1.0, atol=1e-13)
    assert abs(P - Pref).max() < 1e-10
    G.compute_fourier_basis()
    e_ref = np.linalg.eigvalsh(Go.L.toarray())
    assert np.abs(G.e - e_ref).max() < 1e-9 * max(1.0, e_ref[-1])
    assert abs(G.lmax - e_ref[-1]) < 1e-9 * e_ref[-1]  # pygsp: the exact largest eigenvalue once the basis exists
    U = G.U
    assert U.shape == (900, 900) and np.abs(U.T @ U - np.eye(900)).max() < 1e-9
    L = Go.L.toarray()
    assert np.abs(L @ U - U * G.e[None, :]).max() < 1e-8 * e_ref[-1]  # eigenvectors of the ORACLE's Laplacian, caller's cell order
    with pytest.raises(NotImplementedError):
        meld.MELD(knn=15, verbose=0).fit(rng.normal(size=(17000, 5))).graph.compute_fourier_basis()


def test_lmax_method_is_remembered():
    """An estimate made by one method is not silently kept when the other one is asked for (and an injected value is)."""
    meld = _meld()
    rng = np.random.default_rng(6)
    X = rng.normal(size=(4000, 8))
    labels = rng.integers(0, 2, 4000)
    op = meld.MELD(knn=10, verbose=0)
    op.fit_transform(X, labels)
    G = op.graph
    lm_lanczos = G.lmax
    assert G.lmax_info["method"] == "lanczos"
    op.set_params(lmax="arpack")
    op.transform(labels)
    assert G.lmax_info["method"] == "arpack" and abs(G.lmax - lm_lanczos) < 2e-2 * lm_lanczos
    lm_arpack = G.lmax
    G.estimate_lmax(method="lanczos")  # asked for explicitly: computed again
    assert G.lmax_info["method"] == "lanczos" and abs(G.lmax - lm_lanczos) < 1e-9 * lm_lanczos
    op.set_params(lmax=3.25)
    op.transform(labels)
    assert G.lmax == 3.25 and G.lmax_info["method"] == "injected"
    G.estimate_lmax(method="arpack")  # an injected value stays (pygsp's no-op)
    assert G.lmax == 3.25
    op.set_params(lmax=None)
    op.transform(labels)
    assert G.lmax_info["method"] == "lanczos" and abs(G.lmax - lm_lanczos) < 1e-9 * lm_lanczos
    assert lm_arpack > 0


@pytest.mark.parametrize("n", [4000, 90000])  # CSR-stream kernel / panel-tiled layout
def test_one_reduction_lanczos_equals_the_device_resident_loop(n):
    """The sharded driver's iteration with one all-reduce (meld_lanczos_fold / meld_lanczos_axpy3: un-normalised iterate,
    beta known one iteration late) stops at the same prefix with the same Ritz value as the single-GPU loop and as the
    two-all-reduce phases."""
    meld = _meld()
    import torch

    from meld_amd import filter as mf

    rng = np.random.default_rng(8)
    X = rng.normal(size=(n, 6))
    G = meld.MELD(knn=10, verbose=0).fit(X).graph
    ops = G.ops
    idx = torch.arange(G.n_pad, dtype=torch.float64, device=G.val.device)
    u = torch.frac(torch.sin(idx * 12.9898 + 1.0) * 43758.5453) - 0.5
    t_dev, i_dev = mf._lanczos_lmax_device(G, ops, u, 1e-3, 300, 5)
    t_pha, i_pha = mf._lanczos_lmax_phases(G, ops, None, u, 1e-3, 300, 5)
    t_fld, i_fld = mf._lanczos_lmax_folded(G, ops, None, u, 1e-3, 300, 5)
    assert i_fld["iterations"] == i_dev["iterations"] == i_pha["iterations"] and i_fld["all_reduces_per_iteration"] == 1
    assert abs(t_fld - t_dev) < 1e-10 * t_dev and abs(t_pha - t_dev) < 1e-10 * t_dev


@pytest.mark.parametrize("n", [4000, 90000])  # CSR-stream kernel (its batches run out) / panel-tiled layout (stop flag)
def test_overlapped_convergence_checks_stop_where_the_serial_loop_stops(n, monkeypatch):
    """The lmax estimate checks a batch on the host while the next one runs and voids the rest of it once a check has passed:
    same prefix examined, same Ritz value as the loop that waits for every batch; a second estimate right behind a voided batch starts clean."""
    meld = _meld()
    import torch

    from meld_amd import filter as mf

    rng = np.random.default_rng(9)
    X = rng.normal(size=(n, 6))
    G = meld.MELD(knn=10, verbose=0).fit(X).graph
    idx = torch.arange(G.n_pad, dtype=torch.float64, device=G.val.device)
    u = torch.frac(torch.sin(idx * 12.9898 + 1.0) * 43758.5453) - 0.5
    monkeypatch.setenv("MELD_LANCZOS_SPECULATE", "0")
    t_ser, i_ser = mf._lanczos_lmax_device(G, G.ops, u, 1e-3, 300, 5)
    monkeypatch.setenv("MELD_LANCZOS_SPECULATE", "1")
    t_ovl, i_ovl = mf._lanczos_lmax_device(G, G.ops, u, 1e-3, 300, 5)
    t_again, i_again = mf._lanczos_lmax_device(G, G.ops, u, 1e-3, 300, 5)
    torch.cuda.synchronize()
    assert i_ser["iterations"] == i_ovl["iterations"] == i_again["iterations"]
    # (the partial sums of a Lanczos step are added into their slots by atomics: equal up to the order of those additions)
    assert abs(t_ser - t_ovl) <= 1e-12 * t_ser and abs(t_again - t_ovl) <= 1e-12 * t_ser
    assert i_ovl["enqueued"] >= i_ovl["iterations"] and i_ser["enqueued"] <= i_ser["iterations"] + 4
    # a tight iteration cap: the last batch is the cap's, speculation does not run past it
    t_cap, i_cap = mf._lanczos_lmax_device(G, G.ops, u, 1e-12, 23, 5)
    assert i_cap["iterations"] == 23 and i_cap["enqueued"] == 23 and t_cap > 0


def test_results_lent_from_pinned_buffers_stay_valid_and_come_back():
    """The densities are handed over in the pinned buffer they left the device through (no second host copy).  A result that
    is still held must not be touched by later calls -- more results than the pool lends out are copied as before -- and a
    released result returns its buffer to the pool."""
    import gc

    import torch

    meld = _meld()
    from meld_amd import filter as mf

    rng = np.random.default_rng(12)
    held = []
    for i in range(mf._PinnedPool.MAX_OUT + 2):
        X = rng.normal(size=(9000, 6))
        labels = rng.integers(0, 2, 9000)
        out = meld.MELD(knn=7, verbose=0).fit_transform(X, labels)
        held.append((out, out.values.copy()))
    for out, snap in held:
        np.testing.assert_array_equal(out.values, snap)
    key = ((9000, 2), torch.float64)
    assert mf._POOL.out.get(key, 0) == mf._PinnedPool.MAX_OUT
    del held, out
    gc.collect()
    assert mf._POOL.out.get(key, 0) == 0 and len(mf._POOL.free.get(key, [])) == mf._PinnedPool.MAX_OUT


@pytest.mark.gpu
@pytest.mark.parametrize("d", [1, 7, 50, 141])
def test_row_gather_equals_index_select(d):
    """meld_gather_rows_f64 (the cells brought into the device order): even and odd row lengths, a permutation and repeats."""
    import torch

    from meld_amd.graph import HipOps

    g = torch.Generator(device="cuda").manual_seed(d)
    X = torch.randn(5003, d, dtype=torch.float64, device="cuda", generator=g)
    for perm in (torch.randperm(5003, device="cuda", generator=g), torch.randint(0, 5003, (777,), device="cuda", generator=g)):
        assert torch.equal(HipOps().gather_rows(X, perm), X.index_select(0, perm))
