"""``Benchmarker`` (reference ``meld/benchmark.py:10-204``): random ground-truth signals over a dataset,
sample labels drawn from them, MELD likelihoods and their MSE -- the accuracy harness of the
reference's parameter searches.  Host-side NumPy like the reference, with the graph
(``fit_graph``) and the MELD run (``calculate_MELD_likelihood``) on the device.  ``fit_phate`` needs the
optional ``phate`` package exactly as in the reference; every other method works from a supplied 3-D
embedding (``set_phate``).
"""
from __future__ import annotations

import numpy as np
import scipy.special
import scipy.stats

__all__ = ["Benchmarker"]


class Benchmarker(object):
    """Creates random signals over a dataset for benchmarking (reference ``meld/benchmark.py:10-57``;
    same attributes)."""

    def __init__(self, seed=None):
        self.seed = seed
        self.data_phate = None
        self.pdf = None
        self.sample_indicator = None
        self.sample_labels = None
        self.graph = None
        self.graph_kNN = None
        self.meld_op = None
        self.sample_densities = None
        self.estimates = {}

    def set_seed(self, seed):
        self.seed = seed
        return self.seed

    def set_phate(self, data_phate):
        """Stores the (mean-centred, else z-scored) 3-D embedding (reference ``benchmark.py:74-94``)."""
        if not data_phate.shape[1] == 3:
            raise ValueError("data_phate must have 3 dimensions")
        if not np.isclose(data_phate.mean(), 0):
            data_phate = scipy.stats.zscore(data_phate, axis=0)
        self.data_phate = data_phate

    def fit_graph(self, data, n_pca=100, **kwargs):
        """``graphtools.Graph(data, n_pca=n_pca, use_pygsp=True, random_state=seed, **kwargs)`` of reference
        ``benchmark.py:96-113`` -- graphtools' own defaults, i.e. knn=5, decay=40, thresh=1e-4 and
        anisotropy=0 (MELD itself passes anisotropy=1) -- built on the device."""
        import torch

        from .graph import build_knn_graph
        from .pca import pca_project

        opts = dict(knn=5, decay=40, thresh=1e-4, anisotropy=0)
        unknown = [k for k in kwargs if k not in opts and k not in ("n_jobs", "verbose")]
        if unknown:
            raise NotImplementedError("graph options {} are not implemented by the MI355X graph builder".format(sorted(unknown)))
        opts.update({k: v for k, v in kwargs.items() if k in opts})
        X = torch.from_numpy(np.ascontiguousarray(np.asarray(getattr(data, "values", data)), dtype=np.float64)).to("cuda")
        if n_pca is not None and n_pca < min(tuple(X.shape)):
            X = pca_project(X, n_pca, seed=42 if self.seed is None or not isinstance(self.seed, (int, np.integer)) else int(self.seed))
        self.graph = build_knn_graph(X, **opts)
        return self.graph

    def fit_phate(self, data, **kwargs):
        """3-D PHATE embedding of the data (reference ``benchmark.py:115-134``); needs ``phate``."""
        import phate  # optional dependency, as in the reference

        self.set_phate(phate.PHATE(n_components=3, **kwargs).fit_transform(data))
        return self.data_phate

    def generate_ground_truth_pdf(self, data_phate=None):
        """Random convex combination of the embedding's axes through a logistic (reference
        ``benchmark.py:136-175``; same stream of ``np.random`` draws)."""
        np.random.seed(self.seed)
        if data_phate is not None:
            self.set_phate(data_phate)
        elif self.data_phate is None:
            raise ValueError("data_phate must be set prior to running generate_ground_truth_pdf().")
        data_simplex = np.sort(np.random.uniform(size=(2)))
        data_simplex = np.hstack([0, data_simplex, 1])
        data_simplex = np.diff(data_simplex)
        np.random.shuffle(data_simplex)
        sort_axis = np.sum(self.data_phate * data_simplex, axis=1)
        self.pdf = scipy.special.expit(sort_axis)
        return self.pdf

    def generate_sample_labels(self):
        """Bernoulli(pdf) sample labels (reference ``benchmark.py:177-184``)."""
        np.random.seed(self.seed)
        self.sample_indicator = np.random.binomial(1, self.pdf)
        self.sample_labels = np.array(["ctrl" if ind == 0 else "expt" for ind in self.sample_indicator])

    def calculate_MELD_likelihood(self, data=None, **kwargs):
        """MELD likelihood of the "expt" condition on the fitted graph (reference ``benchmark.py:186-200``)."""
        from . import MELD, utils

        np.random.seed(self.seed)
        if not self.graph:
            if data is not None:
                self.fit_graph(data)
            else:
                raise NameError("Must pass `data` unless graph has already been fit")
        self.meld_op = MELD(**kwargs, verbose=False).fit(self.graph)
        self.sample_densities = self.meld_op.transform(self.sample_labels)
        self.sample_likelihoods = utils.normalize_densities(self.sample_densities)
        self.expt_likelihood = self.sample_likelihoods["expt"].values  # only the expt condition
        return self.expt_likelihood

    def calculate_mse(self, estimate):
        """MSE between the ground-truth pdf and an estimate (reference ``benchmark.py:202-204``)."""
        return float(np.mean((np.asarray(self.pdf, dtype=np.float64) - np.asarray(estimate, dtype=np.float64)) ** 2))
