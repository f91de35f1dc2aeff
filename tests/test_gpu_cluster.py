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
