"""Host-side contract of the estimator that needs no GPU: constructor validation, indicator
construction, parameter / reset semantics, type guards.  Mirrors reference test/test_meld.py
(:17-28, :96-105, :175-181) and the descriptor semantics of meld/meld.py:42-141."""
import numpy as np
import pandas as pd
import pytest

import meld_amd as meld


def test_exports_match_reference_names():
    for name in ("MELD", "get_meld_cmap", "normalize_densities", "utils", "__version__"):
        assert hasattr(meld, name)
    assert meld.VertexFrequencyCluster.__name__ == "VertexFrequencyCluster"
    assert meld.Benchmarker.__name__ == "Benchmarker"  # host-side helper, reference meld/__init__.py:3


def test_constructor_defaults():
    op = meld.MELD()
    assert (op.beta, op.offset, op.order, op.filter, op.solver, op.chebyshev_order) == (60, 0, 1, "heat", "chebyshev", 50)
    assert (op.lap_type, op.sample_normalize, op.anisotropy, op.n_landmark) == ("combinatorial", True, 1, None)
    assert (op.knn, op.decay, op.n_pca, op.thresh, op.distance) == (5, 40, 100, 1e-4, "euclidean")
    assert op.graph is None and op.sample_densities is None


def test_invalid_lap_type_message():
    lap_type = "hello world"
    with pytest.raises(ValueError) as e:
        meld.MELD(verbose=0, lap_type=lap_type)
    assert str(e.value) == (
        "lap_type value {} not recognized. " "Choose from ['combinatorial', 'normalized']".format(lap_type)
    )


@pytest.mark.parametrize(
    "kw,msg",
    [
        (dict(beta=-1), "Expected beta > 0, got -1"),
        (dict(filter="gauss"), "filter value gauss not recognized. Choose from ['heat', 'laplacian']"),
        (dict(solver="cg"), "solver value cg not recognized. Choose from ['chebyshev', 'exact']"),
        (dict(chebyshev_order=2.5), "Expected chebyshev_order integer, got 2.5"),
        (dict(chebyshev_order=0), "Expected chebyshev_order > 0, got 0"),
        (dict(knn=0), "Expected knn > 0, got 0"),
        (dict(decay=-3), "Expected decay > 0, got -3"),
    ],
)
def test_validation_messages(kw, msg):
    with pytest.raises(ValueError) as e:
        meld.MELD(**kw)
    assert str(e.value) == msg


def test_check_graph_type_guard():
    with pytest.raises(TypeError) as e:
        meld.utils._check_pygsp_graph(G="hello world")
    assert str(e.value) == (
        "Input graph should be of type graphtools.base.BaseGraph. "
        "With graphtools, use the `use_pygsp=True` flag."
    )
    with pytest.raises(TypeError):
        meld.MELD().transform(np.array(["a", "b"]))  # transform before fit


def test_sample_labels_2d_message():
    labels = np.ones((10, 2))
    with pytest.raises(ValueError) as e:
        meld.MELD()._create_sample_indicators(labels)
    assert str(e.value) == "sample_labels must be a single column. Got" "shape={}".format(labels.shape)


def test_indicator_columns_sorted_and_binary():
    op = meld.MELD()
    ind = op._create_sample_indicators(np.array(["B", "A", "C", "A", "B"]))
    assert list(ind.columns) == ["A", "B", "C"] and list(op.samples) == ["A", "B", "C"]
    assert ind.values.tolist() == [[0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 0, 0], [0, 1, 0]]
    # numeric labels and column-vector DataFrame input
    df = pd.DataFrame(np.array([[1.0], [0.0], [1.0]]), index=["x", "y", "z"], columns=["lab"])
    op._labels_index = df.index
    ind = op._create_sample_indicators(df)
    assert list(ind.columns) == [0.0, 1.0] and list(ind.index) == ["x", "y", "z"]
    assert ind.values.tolist() == [[0, 1], [1, 0], [0, 1]]


def test_set_params_semantics_without_graph():
    op = meld.MELD()
    op.sample_densities = "sentinel"
    op.set_params(beta=op.beta)  # unchanged value: nothing is reset
    assert op.sample_densities == "sentinel"
    op.set_params(beta=op.beta + 1)
    assert op.sample_densities is None and op.beta == 61
    op._graph = "sentinel-graph"
    op.set_params(verbose=1)  # passive parameter keeps the graph
    assert op.graph == "sentinel-graph"
    op.set_params(knn=op.knn + 1)  # graph parameter drops it (reference test/test_meld.py:90-93)
    assert op.graph is None and op.sample_densities is None
    with pytest.raises(ValueError):
        op.set_params(not_a_parameter=1)


def test_unknown_filter_raises_not_implemented():
    from meld_amd.filter import spectral_kernel

    with pytest.raises(NotImplementedError):
        spectral_kernel("gaussian", 1, 0, 1, 1.0)


def test_chebyshev_coefficients_reproduce_the_kernel():
    from meld_amd.filter import chebyshev_coefficients, spectral_kernel

    lmax = 0.37
    h = spectral_kernel("heat", 60, 0, 1, lmax)
    c = chebyshev_coefficients(h, lmax, 50)
    x = np.linspace(0, lmax, 101)
    t = (x - lmax / 2) / (lmax / 2)
    T = np.polynomial.chebyshev.chebval(t, np.concatenate([[c[0] / 2], c[1:]]))
    assert np.abs(T - h(x)).max() < 1e-6


def test_fit_without_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        meld.MELD().fit(np.random.normal(size=(50, 3)))


def test_bench_workload_generator_is_the_oracles():
    """bench.py restates the synthetic workload so that its timed path never imports oracle/; the
    cpu_baseline leg uses the oracle's generator -- both must produce the same cells and labels."""
    import importlib.util
    import os

    from oracle import meld_oracle as mo

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for n, d in ((500, 50), (300, 6)):
        Xa, la = bench.synthetic_cells(n, n_dims=d, seed=3)
        Xb, lb = mo.synthetic_cells(n, n_dims=d, seed=3)
        assert np.array_equal(Xa, Xb) and np.array_equal(la, lb)


def test_vertex_frequency_cluster_argument_checks():
    """Same checks and messages as reference meld/cluster.py (test/test_meld.py:296-345); they all fire
    before any device work."""
    vfc = meld.VertexFrequencyCluster(window_sizes=np.array([2, 4, 8, 24]))
    assert vfc.window_count == 4 and vfc.n_clusters == 10 and vfc._sklearn_params == {"n_init": 10}
    assert list(meld.VertexFrequencyCluster(window_count=5).window_sizes) == [1, 2, 4, 8, 16]
    with pytest.raises(ValueError, match="Estimator must be `fit` before running `transform`."):
        vfc.transform(sample_indicator=np.zeros(5))
    with pytest.raises(ValueError, match=r"Estimator is not fit. Call VertexFrequencyCluster.fit\(\)."):
        vfc.predict()
    vfc.isfit, vfc.N = True, 5  # what fit() records
    with pytest.raises(TypeError, match="`sample_indicator` must be array-like"):
        vfc.transform(sample_indicator="invalid")
    with pytest.raises(TypeError, match="`likelihood` must be array-like"):
        vfc.transform(sample_indicator=np.zeros(5), likelihood="invalid")
    with pytest.raises(ValueError, match="At least one axis of `sample_indicator` must be of length `N`."):
        vfc.transform(sample_indicator=np.zeros(4))
    with pytest.raises(ValueError, match="must have the same shape"):
        vfc.transform(sample_indicator=np.zeros(5), likelihood=np.zeros((5, 2)))
    with pytest.raises(ValueError, match=r"Estimator is not transformed. Call VertexFrequencyCluster.transform\(\)."):
        vfc.predict()
    vfc.set_kmeans_params(n_clusters=4, n_init=3)
    assert vfc.n_clusters == 4 and vfc._sklearn_params == {"n_init": 3}


# ---- Benchmarker (reference meld/benchmark.py, test/test_benchmark.py:34-58): host-side parts ------------------
def test_benchmarker_messages():
    b = meld.Benchmarker()
    assert b.set_seed(0) == 0
    with pytest.raises(ValueError, match="data_phate must have 3 dimensions"):
        b.set_phate(np.random.normal(0, 2, (10, 2)))
    with pytest.raises(ValueError, match=r"data_phate must be set prior to running generate_ground_truth_pdf\(\)."):
        meld.Benchmarker().generate_ground_truth_pdf()
    with pytest.raises(NameError, match="Must pass `data` unless graph has already been fit"):
        meld.Benchmarker().calculate_MELD_likelihood()


def test_benchmarker_ground_truth_follows_the_reference_draws():
    """Same seed -> same pdf and labels as the reference's sequence of global-RNG draws (meld/benchmark.py:154-184),
    restated here with scipy's zscore / expit."""
    import scipy.special
    import scipy.stats

    rng = np.random.default_rng(3)
    emb = rng.normal(1.0, 2.0, (500, 3))  # not centred: goes through the z-score
    b = meld.Benchmarker(seed=7)
    pdf = b.generate_ground_truth_pdf(emb)
    b.generate_sample_labels()

    z = scipy.stats.zscore(emb, axis=0)
    np.random.seed(7)
    w = np.sort(np.random.uniform(size=(2)))
    w = np.diff(np.hstack([0, w, 1]))
    np.random.shuffle(w)
    want = scipy.special.expit(np.sum(z * w, axis=1))
    np.testing.assert_allclose(pdf, want, rtol=1e-13)
    np.random.seed(7)
    ind = np.random.binomial(1, want)
    assert np.array_equal(b.sample_indicator, ind)
    assert np.array_equal(b.sample_labels, np.where(ind == 0, "ctrl", "expt"))
    assert b.calculate_mse(want) < 1e-26
    # an already centred embedding is kept as it is; one passed again replaces the stored one
    b.set_phate(z)
    assert b.data_phate is not None and np.allclose(b.data_phate, z)
    b.set_phate(b.data_phate + 1)
    np.testing.assert_allclose(b.data_phate.mean(axis=0), 0, atol=1e-12)


def test_metric_front_end_reduces_to_the_euclidean_search_or_refuses():
    """distance= enters through the data (meld_amd.graph.metric_front_end): unit rows and a doubled decay for cosine / correlation,
    a doubled decay for sqeuclidean, nothing for euclidean / l2; zero (constant) rows and other metrics are refused by name."""
    import torch

    from meld_amd.graph import metric_front_end

    rng = np.random.default_rng(0)
    X = torch.from_numpy(rng.normal(size=(50, 6)) + 1.0)
    for name in ("euclidean", "l2"):
        Y, decay, to_metric = metric_front_end(X, name, 40)
        assert Y is X and decay == 40 and to_metric is None
    Y, decay, to_metric = metric_front_end(X, "sqeuclidean", 40)
    assert Y is X and decay == 80 and float(to_metric(torch.tensor(3.0))) == 9.0
    Y, decay, to_metric = metric_front_end(X, "cosine", None)
    assert decay is None and torch.allclose(torch.linalg.vector_norm(Y, dim=1), torch.ones(50, dtype=torch.float64))
    # cosine distance of two rows = half the squared distance of their unit rows
    d_cos = 1.0 - float((X[0] @ X[1]) / (torch.linalg.vector_norm(X[0]) * torch.linalg.vector_norm(X[1])))
    assert abs(float(to_metric(torch.linalg.vector_norm(Y[0] - Y[1]))) - d_cos) < 1e-14
    Y, decay, _ = metric_front_end(X, "correlation", 10)
    assert decay == 20 and float(Y.sum(dim=1).abs().max()) < 1e-12
    Z = X.clone()
    Z[3] = 0.0
    with pytest.raises(ValueError, match="all-zero"):
        metric_front_end(Z, "cosine", 40)
    Z[3] = 2.5
    with pytest.raises(ValueError, match="constant"):
        metric_front_end(Z, "correlation", 40)
    with pytest.raises(NotImplementedError, match="manhattan"):
        metric_front_end(X, "manhattan", 40)
    assert meld.MELD(distance="manhattan", verbose=0).distance == "manhattan"  # (served densely at small N, meld_amd/dense.py)
    with pytest.raises(ValueError):
        meld.MELD(distance="mahalanobis")


def test_graph_option_validation_on_the_host():
    """kernel_symm / theta codes of the merge kernel and the precomputed distance names are checked before any device work."""
    import meld_amd
    from meld_amd.graph import symm_code

    assert symm_code("+", None) == (0, 0.0) and symm_code("*", None) == (1, 0.0)
    assert symm_code("mnn", None) == (2, 1.0) and symm_code("mnn", 0.25) == (2, 0.25)
    with pytest.raises(ValueError):
        symm_code("mnn", 2.0)
    with pytest.raises(ValueError):
        symm_code("x", None)
    with pytest.raises(NotImplementedError):
        symm_code(None, None)
    for name in ("precomputed", "precomputed_distance", "precomputed_affinity", "cosine", "euclidean"):
        assert meld_amd.MELD(distance=name, verbose=0).distance == name
    with pytest.raises(ValueError):
        meld_amd.MELD(distance="precomputed_nonsense")


def test_development_switches_need_meld_dev(monkeypatch):
    """meld_amd/_options.py: the eight public options are read from the environment as they are; every other MELD_* switch keeps its
    default unless MELD_DEV=1 (a stray variable in a production environment cannot change which kernels run)."""
    from meld_amd import _options as mo

    monkeypatch.setenv("MELD_KNN_PRUNE", "0")
    monkeypatch.setenv("MELD_SPMM", "csr")
    monkeypatch.setenv("MELD_DEV", "0")
    assert mo.opt("MELD_KNN_PRUNE", "1") == "1" and not mo.is_set("MELD_KNN_PRUNE")
    assert mo.opt("MELD_SPMM", "auto") == "csr"  # (public)
    monkeypatch.setenv("MELD_DEV", "1")
    assert mo.opt("MELD_KNN_PRUNE", "1") == "0" and mo.is_set("MELD_KNN_PRUNE")
    assert len(mo.PUBLIC) <= 8
    # no module of the package reads a MELD_* variable behind the options module's back
    import os
    import re

    root = os.path.dirname(os.path.abspath(mo.__file__))
    for f in sorted(os.listdir(root)):
        if f.endswith(".py") and f != "_options.py":
            txt = open(os.path.join(root, f)).read()
            assert not re.search(r"os\.environ(\.get\(|\[)\s*\"MELD_", txt), f
    for f in sorted(os.listdir(os.path.join(root, "csrc"))):
        txt = open(os.path.join(root, "csrc", f)).read()
        assert not re.search(r"(?<![_a-z])getenv\(\"MELD_(?!DEV\")", txt), f
