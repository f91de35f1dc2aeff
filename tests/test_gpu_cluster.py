"""VertexFrequencyCluster: the device version against the oracle's
restatement of reference meld/cluster.py at the reference's own test size (600 cells)."""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import meld_amd
    from oracle import meld_oracle as mo

    # one connected blob: the spectrogram is defined up to rotations inside degenerate eigenspaces (the
    # reference inherits LAPACK's choice there), e.g. lambda = 0 of a graph with several components
    rng = np.random.default_rng(9)
    X = rng.normal(size=(600, 5)) * np.array([3.0, 2.0, 1.5, 1.0, 0.5])
    labels = np.where(X[:, 0] + 0.7 * rng.normal(size=600) > 0, "expt", "ctrl")
    op = meld_amd.MELD(knn=7, verbose=0)
    dens = op.fit_transform(X, labels)
    lik = meld_amd.utils.normalize_densities(dens)
    Go = mo.build_graph(X, knn=7)
    from scipy.sparse.csgraph import connected_components

    assert connected_components(Go.W)[0] == 1
    return dict(meld=meld_amd, mo=mo, X=X, labels=labels, op=op, lik=lik, Go=Go, ind=op.sample_indicators)


@pytest.mark.parametrize("window_sizes", [np.array([2, 4, 8, 24]), None])
def test_spectrogram_matches_the_oracle(setup, window_sizes):
    meld, mo = setup["meld"], setup["mo"]
    ind, lik = setup["ind"]["expt"], setup["lik"]["expt"]
    vfc = meld.VertexFrequencyCluster(window_sizes=window_sizes) if window_sizes is not None else meld.VertexFrequencyCluster()
    spec = vfc.fit_transform(setup["op"].graph, sample_indicator=ind, likelihood=lik)
    ref, ref_comb = mo.vfc_transform(setup["Go"].K, setup["Go"].L, ind.values, likelihood=lik.values,
                                     window_sizes=window_sizes)
    assert spec.shape == ref.shape == (600, 600)
    assert np.abs(spec - ref).max() <= 1e-7  # entries are O(1); eigenvector signs cancel in |.|
    assert np.abs(vfc.combined_spectrogram - ref_comb).max() <= 1e-8


def test_two_dimensional_indicators_and_clusters(setup):
    """2-D indicators are concatenated spectrograms (reference test_2d); on well separated structure the
    seeded device KMeans and sklearn's find the same partition, and both are sorted by mean likelihood."""
    meld, mo = setup["meld"], setup["mo"]
    ws = np.array([2, 4, 8, 24])
    vfc = meld.VertexFrequencyCluster(window_sizes=ws, n_clusters=3, random_state=0)
    labels = vfc.fit_predict(setup["op"].graph, sample_indicator=setup["ind"], likelihood=setup["lik"])
    assert len(labels) == 600 and set(labels.tolist()) == {0, 1, 2}
    ref_spec, ref_comb = mo.vfc_transform(setup["Go"].K, setup["Go"].L, setup["ind"].values, likelihood=setup["lik"].values,
                                          window_sizes=ws)
    assert np.abs(vfc.spectrogram - ref_spec).max() <= 1e-7 and vfc.spectrogram.shape == (600, 1200)
    ref_labels = mo.vfc_predict(ref_comb, 3, setup["lik"].values, random_state=0)
    from sklearn.metrics import adjusted_rand_score

    assert adjusted_rand_score(labels, ref_labels) > 0.95
    means = [np.mean(setup["lik"].values[labels == c]) for c in range(3)]
    assert means == sorted(means)
    assert len(vfc.predict(n_clusters=2)) == 600 and vfc.n_clusters == 2  # reference test_predit_setting_n_cluster


def test_no_likelihood_and_size_guard(setup):
    meld = setup["meld"]
    vfc = meld.VertexFrequencyCluster(window_sizes=np.array([2, 4]))
    labels = vfc.fit_predict(setup["op"].graph, sample_indicator=setup["ind"]["expt"])
    assert len(labels) == 600 and vfc.combined_spectrogram is None
    with pytest.raises(ValueError, match="sample_indicator must be 1-dimensional"):
        import torch

        vfc._compute_spectrogram(torch.zeros(600, 2, dtype=torch.float64, device="cuda"), vfc.windows[0])


# ---- the filter-bank method (BASELINE config 5) -----------------------------------

def _exact_heat_window_spectrogram(G, vfc):
    """What the filter-bank method approximates, from the full eigendecomposition: the reference-style spectrogram
    with the method's (polynomial) heat windows, sum_t tanh(|U[j,k]| sqrt(p_t(lambda_k)) / sqrt([p_t(L)]_jj)), and
    the window norms [p_t(L)]_jj."""
    L = np.asarray(G.L.todense())
    lam, U = np.linalg.eigh(L)
    lam = np.clip(lam, 0.0, None)
    T, B = len(vfc.window_sizes), vfc.n_bands
    P = np.polynomial.chebyshev.chebval(2.0 * lam / vfc._fb["lmax"] - 1.0, vfc._fb["coeffs"].T)  # [T*B, N]: p_tb(lambda_k)
    assert P.min() > -1e-9  # Jackson damping keeps the filters non-negative
    pt = np.clip(P.reshape(T, B, -1).sum(1), 0.0, None)  # [T, N]
    norm2 = (U * U) @ pt.T  # [N, T]
    spec = np.zeros_like(U)
    for ti in range(T):
        spec += np.tanh(np.abs(U) * np.sqrt(pt[ti])[None, :] / np.sqrt(norm2[:, ti])[:, None])
    return spec, norm2, lam


def test_filterbank_spectrogram_against_the_full_eigendecomposition(setup):
    """(i) With n_probes = N the Ritz pairs are the eigenpairs: the first N columns equal the reference-style
    spectrogram with heat windows computed from numpy's eigendecomposition, the band columns vanish.  (ii) With 64
    probes: the window norms (deflated Hutchinson estimate) are within a few % of the exact diagonal, the low Ritz
    values are the low eigenvalues, and the seeded result is reproducible bit for bit."""
    meld = setup["meld"]
    G = setup["op"].graph
    full = meld.VertexFrequencyCluster(method="filterbank", n_probes=600, n_bands=12, random_state=3).fit(G)
    assert full.method_ == "filterbank" and full._fb["order"] >= 64
    est = full.transform(setup["ind"]["expt"])
    ref, norm2, lam = _exact_heat_window_spectrogram(G, full)
    assert est.shape == (600, 612)
    assert np.abs(np.sort(est[:, :600], axis=1) - np.sort(ref, axis=1)).max() <= 1e-6  # (column order: by Ritz value)
    assert np.abs(est[:, 600:]).max() <= 1e-6
    vfc = meld.VertexFrequencyCluster(method="filterbank", n_probes=64, n_bands=12, random_state=3).fit(G)
    got = vfc._fb["window_norm2"].cpu().numpy()
    rel = np.abs(got - norm2) / norm2
    print("filter bank, 64 probes: window norms rel. error mean %.4f max %.4f" % (rel.mean(), rel.max()))
    assert rel.mean() <= 0.05 and rel.max() <= 0.6
    th = vfc._fb["ritz"].cpu().numpy()
    assert np.abs(th[:10] - lam[:10]).max() <= 1e-3 * lam[10]
    a = vfc.transform(setup["ind"]["expt"])
    b = meld.VertexFrequencyCluster(method="filterbank", n_probes=64, n_bands=12, random_state=3).fit_transform(G, setup["ind"]["expt"])
    assert a.shape == (600, 76) and np.array_equal(a, b)
    # the indicator enters only through its zero pattern, like the reference's spectrogram (oracle test below)
    ind0 = setup["ind"]["expt"].values.astype(float).copy()
    spec0 = vfc.transform(ind0, center=False)
    assert np.all(spec0[ind0 == 0] == 0) and np.array_equal(spec0[ind0 != 0], a[ind0 != 0])


def test_reference_spectrogram_is_independent_of_the_signal(setup):
    """Why the filter-bank method may ignore the indicator's values: in the reference (restated by the oracle) the
    indicator multiplies whole columns, which are then l2-normalised and taken in absolute value."""
    mo, Go = setup["mo"], setup["Go"]
    rng = np.random.default_rng(0)
    a, _ = mo.vfc_transform(Go.K, Go.L, setup["ind"]["expt"].values.astype(float), window_sizes=np.array([1, 2, 4]))
    b, _ = mo.vfc_transform(Go.K, Go.L, rng.normal(size=600), window_sizes=np.array([1, 2, 4]))
    assert np.abs(a - b).max() <= 1e-12


def test_filterbank_clusters_like_the_reference_on_clear_structure():
    """Small-N limit in clustering terms: three groups of cells of different local density and connectivity; the
    filter-bank method's clusters agree with the reference algorithm's (dense method == oracle) clusters."""
    import meld_amd
    from sklearn.metrics import adjusted_rand_score

    rng = np.random.default_rng(5)
    X = np.concatenate([rng.normal(0.0, 0.35, size=(400, 4)), rng.normal(0.0, 1.0, size=(400, 4)) + np.array([6.0, 0, 0, 0]),
                        rng.normal(0.0, 2.5, size=(400, 4)) + np.array([0, 14.0, 0, 0])])
    X += 0.01 * rng.normal(size=X.shape)
    labels = np.where(rng.random(1200) < 0.5, "expt", "ctrl")
    op = meld_amd.MELD(knn=10, verbose=0)
    lik = meld_amd.utils.normalize_densities(op.fit_transform(X, labels))
    dense = meld_amd.VertexFrequencyCluster(n_clusters=3, random_state=0, method="dense")
    ld = dense.fit_predict(op.graph, sample_indicator=op.sample_indicators["expt"], likelihood=lik["expt"])
    fb = meld_amd.VertexFrequencyCluster(n_clusters=3, random_state=0, method="filterbank", n_probes=256)
    lf = fb.fit_predict(op.graph, sample_indicator=op.sample_indicators["expt"], likelihood=lik["expt"])
    truth = np.repeat([0, 1, 2], 400)
    assert adjusted_rand_score(ld, truth) > 0.9  # the structure is clear to the reference algorithm ...
    assert adjusted_rand_score(lf, truth) > 0.9  # ... and to the filter bank
    assert adjusted_rand_score(lf, ld) > 0.9


def test_kmeans_kernel_equals_a_plain_lloyd_step():
    import torch
    from meld_amd.cluster import _kmeans, _lloyd_step
    from meld_amd._lib import get_lib

    rng = np.random.default_rng(2)
    Y = torch.from_numpy(rng.normal(size=(70_001, 7))).cuda()
    C = Y[:5].clone().contiguous()
    nb = 64
    scratch = dict(nb=nb, sum=torch.empty(nb * 5 * 7, dtype=torch.float64, device="cuda"), cnt=torch.empty(nb * 5, dtype=torch.float64, device="cuda"),
                   **{"in": torch.empty(nb, dtype=torch.float64, device="cuda")})
    lab = torch.empty(70_001, dtype=torch.int32, device="cuda")
    newC, inertia = _lloyd_step(Y, C, lab, scratch)
    D = torch.cdist(Y, C) ** 2
    ref_lab = D.argmin(1)
    assert torch.equal(lab.to(torch.int64), ref_lab)
    refC = torch.stack([Y[ref_lab == j].mean(0) for j in range(5)])
    assert torch.allclose(newC, refC, rtol=1e-12, atol=1e-12)
    assert abs(float(inertia) - float(D.gather(1, ref_lab[:, None]).sum())) <= 1e-9 * float(inertia)
    out = _kmeans(Y, 4, n_init=2, seed=1)
    assert out.shape == (70_001,) and set(out.unique().tolist()) == {0, 1, 2, 3}
    # beyond the kernel's LDS staging (d > 32 or k > 64): the library Lloyd step, same outputs
    from meld_amd.cluster import _lloyd_step_library

    lab2 = torch.empty_like(lab)
    newC2, inertia2 = _lloyd_step_library(Y, C, lab2)
    assert torch.equal(lab2, lab) and torch.allclose(newC2, newC, rtol=1e-12, atol=1e-12)
    assert abs(float(inertia2) - float(inertia)) <= 1e-9 * float(inertia)
    Yw = torch.from_numpy(rng.normal(size=(5000, 40)) + 6.0 * rng.integers(0, 2, size=(5000, 1))).cuda()
    out = _kmeans(Yw, 3, n_init=2, seed=0)
    assert out.shape == (5000,) and set(out.unique().tolist()) == {0, 1, 2}


def test_more_clusters_than_the_kmeans_kernel_stages(setup):
    """n_clusters = 40 (reference meld/cluster.py:340-345 has no limit: PCA to n_clusters features, KMeans): the
    device KMeans falls back to the library Lloyd step beyond 32 features / 64 clusters."""
    meld = setup["meld"]
    vfc = meld.VertexFrequencyCluster(n_clusters=40, random_state=0, window_sizes=np.array([2, 4, 8, 24]))
    labels = vfc.fit_predict(setup["op"].graph, sample_indicator=setup["ind"]["expt"], likelihood=setup["lik"]["expt"])
    assert labels.shape == (600,) and len(np.unique(labels)) > 30 and labels.min() == 0 and labels.max() <= 39


def test_filterbank_at_a_size_the_dense_method_cannot_reach():
    """300k cells (the dense reference algorithm would need a 720 GB Fourier basis): fit / transform / predict run
    on the hot path's recurrence, auto-selected; the clusters are sorted by mean likelihood."""
    import meld_amd
    from bench import synthetic_cells

    X, labels = synthetic_cells(300_000, 50, seed=1)
    op = meld_amd.MELD(knn=15, chebyshev_order=30, verbose=0)
    lik = meld_amd.utils.normalize_densities(op.fit_transform(X, labels))
    vfc = meld_amd.VertexFrequencyCluster(n_clusters=6, random_state=0, n_probes=32, window_sizes=np.array([1, 2, 4, 8, 16, 32]), n_init=2)
    out = vfc.fit_predict(op.graph, sample_indicator=op.sample_indicators["expt"], likelihood=lik["expt"])
    assert vfc.method_ == "filterbank" and vfc.spectrogram.shape == (300_000, 32 + 16)
    assert out.shape == (300_000,) and set(np.unique(out).tolist()) == set(range(6))
    means = [lik["expt"].values[out == c].mean() for c in range(6)]
    assert means == sorted(means)
    assert np.isfinite(vfc.spectrogram).all() and vfc.spectrogram.min() >= 0.0


def test_wide_recurrence_step_equals_the_two_column_kernel():
    """meld_cheby_step_wide (lanes = columns, the matrix streamed once for all columns) against the recurrence kernel MELD's own
    filter uses, column pair by column pair: p = 64 and a ragged p = 41, with and without the z operand, aliased y / z, and on a
    row shard (x_row_offset > 0)."""
    import torch

    import meld_amd
    from meld_amd import filter as mf
    from oracle import meld_oracle as mo

    X, _ = mo.synthetic_cells(70000, n_dims=20, seed=4)
    G = meld_amd.build_knn_graph(torch.from_numpy(X).cuda(), knn=9)
    ops = mf._ops_of(G)
    n = G.N
    gen = torch.Generator(device="cuda").manual_seed(1)
    for p in (64, 41):
        x = torch.rand(n, p, dtype=torch.float64, device="cuda", generator=gen)
        z = torch.rand(n, p, dtype=torch.float64, device="cuda", generator=gen)
        for gamma in (0.0, -1.0):
            y = z.clone()
            ops.cheby_step_wide(G, p, x, 0, y if gamma != 0.0 else None, y, 0.7, -0.2, gamma)
            ref = torch.empty_like(x)
            for c in range(p):  # one column at a time through the product kernel
                xc, zc = x[:, c].contiguous(), z[:, c].contiguous()
                yc = torch.empty_like(xc)
                ops.cheby_step(G, 1, xc, 0, zc if gamma != 0.0 else None, yc, None, 0.7, -0.2, gamma, 0.0)
                ref[:, c] = yc
            assert float((y - ref).abs().max() / ref.abs().max()) < 1e-13, (p, gamma)


def test_filterbank_on_the_wide_kernel_equals_the_pair_path():
    """64 probes: the filter bank on the wide kernel (row-major iterates) and on the two-column kernel (pair-major iterates) are the
    same algorithm on the same random signs."""
    import os

    import torch

    import meld_amd
    from oracle import meld_oracle as mo

    X, labels = mo.synthetic_cells(30000, n_dims=12, seed=8)
    G = meld_amd.MELD(knn=9, verbose=0).fit(torch.from_numpy(X).cuda()).graph
    out = {}
    for w in ("1", "0"):
        os.environ["MELD_VFC_WIDE"] = w
        try:
            vfc = meld_amd.VertexFrequencyCluster(method="filterbank", n_probes=64, n_bands=8, window_sizes=np.array([1, 2, 4, 8]),
                                                  chebyshev_order=48, random_state=3).fit(G)
        finally:
            os.environ.pop("MELD_VFC_WIDE", None)
        assert vfc._fb["spmm"] == ("wide" if w == "1" else "pairs")
        out[w] = vfc._fb_spectrogram.cpu().numpy()
    assert np.abs(out["1"] - out["0"]).max() < 1e-8 * max(1.0, np.abs(out["0"]).max())
