"""Vertex-Frequency Clustering (reference ``meld/cluster.py:13-367``) at small N on the device.

SURVEY.md section 8f row 2 (i): the reference algorithm is dense -- windows are powers of the
diffusion operator (``cluster.py:158-194``), the spectrogram is a windowed graph Fourier transform
over ALL eigenvectors of the Laplacian (``:98-156``, ``:235-236``) -- so it is O(N^3) and cannot run
at the hot path's sizes; this module is the faithful version for N <= ``dense.DENSE_MAX_N`` on dense
PyTorch-ROCm linear algebra (fp64 rocBLAS GEMMs for the window powers and U^T (W .* s), rocSOLVER
``eigh`` for the Fourier basis; library calls, not hand-written kernels), followed by PCA
(``meld_amd.pca``) and a seeded k-means++ / Lloyd KMeans on the device.  The scalable reformulation
(heat-filter windows + a Chebyshev filter bank instead of the full eigenbasis, section 8f row 2 (ii)) is
not implemented.  Same constructor, methods, checks and messages as the reference class.
"""
from __future__ import annotations

import numpy as np
import pandas as pd
import torch

from . import utils

__all__ = ["VertexFrequencyCluster"]


def _l2_normalize_columns(M):
    nrm = torch.linalg.vector_norm(M, dim=0, keepdim=True)
    return M / torch.where(nrm == 0, torch.ones_like(nrm), nrm)


def _kmeans(Y, k, n_init=10, max_iter=300, tol=1e-4, seed=None):
    """Seeded k-means++ initialisation + Lloyd iterations on the device; best inertia of ``n_init``
    runs (sklearn's ``KMeans`` defaults; its relative tolerance is on the centre shift)."""
    n = Y.shape[0]
    gen = torch.Generator(device=Y.device)
    gen.manual_seed(0 if seed is None else int(seed))
    var_tol = tol * float(Y.var(dim=0, unbiased=False).mean())
    best = None
    for _ in range(max(1, int(n_init))):
        # k-means++
        first = int(torch.randint(0, n, (1,), device=Y.device, generator=gen))
        C = [Y[first]]
        d2 = ((Y - C[0]) ** 2).sum(1)
        for _c in range(1, k):
            tot = d2.sum()
            probs = d2 / tot if float(tot) > 0 else torch.full_like(d2, 1.0 / n)
            nxt = int(torch.multinomial(probs, 1, generator=gen))
            C.append(Y[nxt])
            d2 = torch.minimum(d2, ((Y - C[-1]) ** 2).sum(1))
        C = torch.stack(C)
        for _it in range(max_iter):
            D = torch.cdist(Y, C) ** 2
            lab = D.argmin(1)
            newC = torch.zeros_like(C)
            cnt = torch.bincount(lab, minlength=k).to(Y.dtype)
            newC.index_add_(0, lab, Y)
            newC = torch.where(cnt[:, None] > 0, newC / cnt.clamp(min=1)[:, None], C)
            shift = ((newC - C) ** 2).sum()
            C = newC
            if (_it & 7) == 7 and float(shift) <= var_tol:  # one host sync per 8 iterations
                break
        D = torch.cdist(Y, C) ** 2
        lab = D.argmin(1)
        inertia = float(D.gather(1, lab[:, None]).sum())
        if best is None or inertia < best[0]:
            best = (inertia, lab)
    return best[1]


class VertexFrequencyCluster:
    """Performs Vertex Frequency clustering for data given a raw experimental signal and enhanced
    experimental signal (reference ``meld/cluster.py:13-46``; same parameters)."""

    def __init__(self, n_clusters=10, likelihood_bias=1, window_count=9, window_sizes=None, sparse=False,
                 suppress=False, random_state=None, **kwargs):
        self.suppress = suppress
        self.sparse = sparse  # accepted for API compatibility; the device version is dense either way
        self._basewindow = None
        if window_sizes is None:
            self.window_sizes = np.power(2, np.arange(window_count))
        else:
            self.window_sizes = np.asarray(window_sizes)
        self.window_count = np.min(self.window_sizes.shape)
        self.n_clusters = n_clusters
        self.likelihood_bias = likelihood_bias
        self.random_state = random_state
        self.window = None
        self.eigenvectors = None
        self.N = None
        self.spec_hist = None
        self.spectrogram = None
        self.combined_spectrogram = None
        self.isfit = False
        self.likelihood = None
        self.sample_indicator = None
        self._sklearn_params = {"n_init": 10}
        self._sklearn_params.update(kwargs)

    # -- pieces (reference cluster.py:80-205) --------------------------------------------------------
    def _activate(self, x, alpha=1):
        return torch.tanh(alpha * torch.abs(x))

    def _compute_spectrogram(self, sample_indicator, window):
        """normalize(U^T (window .* s), axis=0)^T for one window (reference ``cluster.py:98-137``);
        ``sample_indicator``: 1-D, in the graph's internal cell order, on the device."""
        if sample_indicator.dim() != 1:
            raise ValueError("sample_indicator must be 1-dimensional. Got shape: {}".format(tuple(sample_indicator.shape)))
        C = window * sample_indicator[None, :]
        C = _l2_normalize_columns(self.eigenvectors.T @ C)
        return C.T

    def _compute_multiresolution_spectrogram(self, sample_indicator):
        spec = torch.zeros((self.windows[0].shape[1], self.eigenvectors.shape[1]), dtype=torch.float64,
                           device=self.eigenvectors.device)
        for window in self.windows:
            spec += self._activate(self._compute_spectrogram(sample_indicator, window))
        return spec

    def _compute_window(self, window, t=1):
        return _l2_normalize_columns(torch.linalg.matrix_power(window, int(t))).T

    def _compute_windows(self):
        windows = []
        cur = self._basewindow
        windows.append(_l2_normalize_columns(cur).T)
        for _ in range(len(self.window_sizes) - 1):
            cur = cur @ cur
            windows.append(_l2_normalize_columns(cur).T)
        return windows

    def _combine_spectrogram_likelihood(self, spectrogram, likelihood):
        spectrogram_n = spectrogram / np.linalg.norm(spectrogram)
        ees_n = likelihood / np.linalg.norm(likelihood, ord=2, axis=0)
        ees_n = ees_n * self.likelihood_bias
        return np.c_[spectrogram_n, ees_n]

    # -- fit (reference cluster.py:207-241) -----------------------------------------------------------
    def fit(self, G):
        """Builds the windows (powers of the diffusion operator) and the graph Fourier basis."""
        from .dense import DENSE_MAX_N

        self.graph = utils._check_pygsp_graph(G)
        G = self.graph
        if G.n_rows != G.N:
            raise ValueError("VertexFrequencyCluster needs an unsharded graph")
        if G.N > DENSE_MAX_N:
            raise NotImplementedError(
                "VertexFrequencyCluster is the reference's dense O(N^3) algorithm; N={} exceeds the {} cells it is "
                "offered for (the scalable filter-bank reformulation is not implemented)".format(G.N, DENSE_MAX_N)
            )
        dev, n = G.val.device, G.N
        # dense kernel (diagonal included) and Laplacian in the graph's internal cell order
        row_of = torch.repeat_interleave(torch.arange(n, device=dev), G.rowptr[1:] - G.rowptr[:-1])
        W = torch.zeros(n, n, dtype=torch.float64, device=dev)
        W[row_of, G.col.to(torch.int64)] = G.val
        if getattr(G, "_kdiag", None) is not None:
            kdiag = G._kdiag.to(dev)
        else:
            kdiag = G.kernel_diagonal()[:n]
        K = W + torch.diag(kdiag)
        self._basewindow = K / K.sum(dim=1, keepdim=True)  # graphtools diff_op
        if np.all(np.diff(np.log2(self.window_sizes)) == 1):
            self.windows = self._compute_windows()
        else:
            self.windows = [self._compute_window(self._basewindow, t=t) for t in self.window_sizes]
        L = torch.diag(G.dw_dev[:n]) - W
        _, self.eigenvectors = torch.linalg.eigh(L)  # pygsp compute_fourier_basis
        self.N = n
        self.isfit = True
        return self

    # -- transform (reference cluster.py:243-309) -----------------------------------------------------
    def transform(self, sample_indicator, likelihood=None, center=True):
        self.sample_indicator = sample_indicator
        self.likelihood = likelihood
        if not self.isfit:
            raise ValueError("Estimator must be `fit` before running `transform`.")
        if not isinstance(self.sample_indicator, (list, tuple, np.ndarray, pd.Series, pd.DataFrame)):
            raise TypeError("`sample_indicator` must be array-like.")
        if likelihood is not None and not isinstance(self.likelihood, (list, tuple, np.ndarray, pd.Series, pd.DataFrame)):
            raise TypeError("`likelihood` must be array-like.")
        self.sample_indicator = np.array(self.sample_indicator)
        if self.N not in self.sample_indicator.shape:
            raise ValueError("At least one axis of `sample_indicator` must be" " of length `N`.")
        if likelihood is not None:
            if self.N not in self.likelihood.shape:
                raise ValueError("At least one axis of `likelihood` must be" " of length `N`.")
            if likelihood.shape != sample_indicator.shape:
                raise ValueError(
                    "`sample_indicator` and `likelihood` must have the same shape. "
                    "Got sample_indicator: {} and likelihood: {}".format(str(sample_indicator.shape), str(likelihood.shape))
                )
            self.likelihood = np.array(self.likelihood)
        if center:
            self.sample_indicator = self.sample_indicator - self.sample_indicator.mean()

        dev = self.eigenvectors.device
        perm = getattr(self.graph, "perm", None)  # internal (locality) order -> caller's order

        def to_internal(v):
            t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)).to(dev)
            return t if perm is None else t.index_select(0, perm)

        def to_caller(spec):
            if perm is None:
                return spec.cpu().numpy()
            out = torch.empty_like(spec)
            out[perm] = spec
            return out.cpu().numpy()

        if self.sample_indicator.ndim == 1:
            self.spectrogram = to_caller(self._compute_multiresolution_spectrogram(to_internal(self.sample_indicator)))
        else:
            self.spectrogram = np.hstack([
                to_caller(self._compute_multiresolution_spectrogram(to_internal(self.sample_indicator[:, i])))
                for i in range(self.sample_indicator.shape[1])
            ])
        if self.likelihood is not None:
            self.combined_spectrogram = self._combine_spectrogram_likelihood(self.spectrogram, self.likelihood)
        return self.spectrogram

    def fit_transform(self, G, sample_indicator, likelihood=None, **kwargs):
        self.fit(G, **kwargs)
        return self.transform(sample_indicator, likelihood, **kwargs)

    # -- predict (reference cluster.py:315-357) --------------------------------------------------------
    def predict(self, n_clusters=None, **kwargs):
        if n_clusters is not None:
            self.n_clusters = n_clusters
        if not self.isfit:
            raise ValueError("Estimator is not fit. " "Call VertexFrequencyCluster.fit().")
        if self.spectrogram is None:
            raise ValueError("Estimator is not transformed. " "Call VertexFrequencyCluster.transform().")
        data = self.spectrogram if self.combined_spectrogram is None else self.combined_spectrogram
        from .pca import pca_project

        params = dict(self._sklearn_params)
        params.update(kwargs)
        dev = self.eigenvectors.device
        Y = pca_project(torch.from_numpy(np.ascontiguousarray(data, dtype=np.float64)).to(dev), self.n_clusters)
        lab = _kmeans(Y, self.n_clusters, n_init=params.get("n_init", 10), max_iter=params.get("max_iter", 300),
                      tol=params.get("tol", 1e-4), seed=params.get("random_state", self.random_state)).cpu().numpy()
        values = self.likelihood if self.likelihood is not None else self.sample_indicator
        # scprep.utils.sort_clusters_by_values: clusters relabelled by ascending mean of the values
        values = np.asarray(values, dtype=np.float64)
        uniq = np.unique(lab)
        means = np.array([np.mean(values[lab == c]) for c in uniq])
        remap = {c: i for i, c in enumerate(uniq[np.argsort(means)])}
        self.labels_ = np.array([remap[c] for c in lab])
        return self.labels_

    def fit_predict(self, G, sample_indicator, likelihood=None, **kwargs):
        self.fit_transform(G, sample_indicator, likelihood, **kwargs)
        return self.predict()

    def set_kmeans_params(self, **kwargs):
        k = kwargs.pop("n_clusters", False)
        if k:
            self.n_clusters = k
        self._sklearn_params = kwargs
