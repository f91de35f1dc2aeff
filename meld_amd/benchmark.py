"""``Benchmarker`` -- synthetic ground truth over a dataset, for quantitative comparisons and parameter
searches (public surface of reference ``meld/benchmark.py:10-200``, exported at ``meld/__init__.py:3``).

Host-side helper around the accelerated path, not part of it: the random ground truth (a logistic
function of a random convex combination of three embedding coordinates) and the Bernoulli sample labels
are NumPy; the graph and the density estimate go through ``meld_amd.MELD`` (``fit`` on a graph,
``transform``, ``normalize_densities`` -- the call sequence of reference ``meld/benchmark.py:194-196``).
The random draws follow the reference's order (``np.random.seed(seed)`` before each stage; two uniforms,
one shuffle; one binomial vector), so a seed produces the same pdf and labels as there.
"""
from __future__ import annotations

import numpy as np

__all__ = ["Benchmarker"]


def _standardise(coords):
    """Column-wise z-scores unless the embedding is already centred (the reference's rule)."""
    coords = np.asarray(coords)
    if coords.ndim != 2 or coords.shape[1] != 3:
        raise ValueError("data_phate must have 3 dimensions")
    if np.isclose(coords.mean(), 0):
        return coords
    return (coords - coords.mean(axis=0)) / coords.std(axis=0)


def _random_simplex_weights():
    """Three non-negative weights summing to one: gaps between two sorted uniforms on [0, 1], shuffled."""
    cuts = np.concatenate([[0.0], np.sort(np.random.uniform(size=2)), [1.0]])
    w = np.diff(cuts)
    np.random.shuffle(w)
    return w


class Benchmarker(object):
    """Creates random signals over a dataset for benchmarking.

    Attributes: ``data_phate`` [n, 3] embedding the ground truth is drawn over, ``pdf`` [n] ground-truth
    P(expt | cell), ``sample_indicator`` [n] 0/1 draws, ``sample_labels`` [n] "ctrl"/"expt", ``graph``,
    ``meld_op``, ``sample_densities``, ``sample_likelihoods``, ``expt_likelihood``, ``estimates``.
    """

    def __init__(self, seed=None):
        self.seed = seed
        for name in ("data_phate", "pdf", "sample_indicator", "sample_labels", "graph", "graph_kNN", "meld_op",
                     "sample_densities"):
            setattr(self, name, None)
        self.estimates = {}

    def set_seed(self, seed):
        self.seed = seed
        return self.seed

    def set_phate(self, data_phate):
        self.data_phate = _standardise(data_phate)

    def fit_graph(self, data, n_pca=100, **kwargs):
        """Graph on ``data`` with the MI355X builder (the reference calls ``graphtools.Graph(data, n_pca=n_pca,
        use_pygsp=True, random_state=seed, **kwargs)``; same keyword names: knn, decay, thresh, anisotropy ...)."""
        from .meld import MELD

        kwargs.pop("use_pygsp", None)
        # graphtools accepts more graph keywords than the device builder implements: those are rejected here, by name,
        # instead of being stored and silently ignored (the graph would differ from the reference's for the same call)
        # (n_jobs and lmax do not change the graph: MELD / GraphEstimator take them, graphtools ran fit_graph(data, n_jobs=-1))
        known = {"knn", "decay", "thresh", "ksel", "sample_idx", "n_landmark", "verbose", "distance", "n_jobs", "lmax",
                 "bandwidth", "bandwidth_scale", "knn_max", "kernel_symm", "theta"}
        unknown = sorted(k for k in kwargs if k not in known | {"anisotropy"})
        if unknown:
            raise NotImplementedError("graph options {} are not implemented by the MI355X graph builder".format(unknown))
        seed = self.seed
        if seed is not None and not isinstance(seed, (int, np.integer)):
            # (graphtools also takes a RandomState; the device PCA is seeded by an integer)
            raise TypeError("Benchmarker.seed must be an integer or None to seed the graph's PCA, got {!r}".format(type(seed).__name__))
        builder = MELD(n_pca=n_pca, random_state=seed, anisotropy=kwargs.pop("anisotropy", 0), **kwargs)
        self.graph = builder.fit(data).graph
        return self.graph

    def fit_phate(self, data, **kwargs):
        """3-d PHATE embedding of ``data`` (needs the optional ``phate`` package, as in the reference)."""
        import phate

        self.set_phate(phate.PHATE(n_components=3, **kwargs).fit_transform(data))
        return self.data_phate

    def generate_ground_truth_pdf(self, data_phate=None):
        np.random.seed(self.seed)
        if data_phate is not None:
            self.set_phate(data_phate)
        elif self.data_phate is None:
            raise ValueError("data_phate must be set prior to running generate_ground_truth_pdf().")
        axis = self.data_phate @ _random_simplex_weights()
        self.pdf = 1.0 / (1.0 + np.exp(-axis))
        return self.pdf

    def generate_sample_labels(self):
        np.random.seed(self.seed)
        self.sample_indicator = np.random.binomial(1, self.pdf)
        self.sample_labels = np.where(self.sample_indicator == 0, "ctrl", "expt")

    def calculate_MELD_likelihood(self, data=None, **kwargs):
        from .meld import MELD
        from .utils import normalize_densities

        np.random.seed(self.seed)
        if not self.graph:
            if data is None:
                raise NameError("Must pass `data` unless graph has already been fit")
            self.fit_graph(data)
        self.meld_op = MELD(**kwargs, verbose=False).fit(self.graph)
        self.sample_densities = self.meld_op.transform(self.sample_labels)
        self.sample_likelihoods = normalize_densities(self.sample_densities)
        self.expt_likelihood = self.sample_likelihoods["expt"].values
        return self.expt_likelihood

    def calculate_mse(self, estimate):
        """Mean squared error between the ground-truth pdf and an estimate of it."""
        return float(np.mean((np.asarray(self.pdf) - np.asarray(estimate)) ** 2))
