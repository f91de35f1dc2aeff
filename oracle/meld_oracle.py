"""CPU oracle for the MELD hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  ``meld_amd`` never does (tests/test_no_oracle_in_product.py enforces it).

What this is
------------
A NumPy / SciPy / scikit-learn restatement of the arithmetic behind
``meld.MELD().fit_transform(X, sample_labels)`` (reference v1.0.2).  The reference package is a thin
orchestration layer; the arithmetic lives in two un-vendored third-party dependencies that are *not
installed here and not reachable* (no network):

* ``graphtools>=1.5.0`` (reference ``setup.py:9``, call sites ``meld/meld.py:5,9,118``) --
  kNN search, alpha-decay kernel, symmetrisation, anisotropy, PyGSP weight matrix;
* ``pygsp`` (unpinned, PyPI latest 0.5.1; reference ``setup.py:12``, call sites
  ``meld/filter.py:1,39,56,59``) -- Laplacian, ``estimate_lmax``, Chebyshev coefficients and
  the three-term recurrence, exact (Fourier) filtering.

Each function below restates the *published* algorithm of those libraries (tagged [UPSTREAM]) or
follows the reference tree (cited file:line relative to /root/reference).  It uses the same
primitives the reference stack uses (sklearn NearestNeighbors, scipy.sparse CSR, ARPACK eigsh,
scipy CSR @ dense), so it doubles as the fair CPU baseline of bench.py.

PARITY STATUS: **partially pinned / otherwise unpinned**.  The reference cannot be imported in the
build container, so golden vectors could not be generated from it.  The oracle is pinned against
the only known-answer test on this path that the reference holds:
``test/test_meld.py:43-81`` (sum of the "treat" density == 532 for both filters, exact solver,
thresh=0 dense graph) -- replayed in tests/test_oracle.py.  The second numeric pin
(``test/test_benchmark.py:23``) needs ``phate`` and is not reproducible.  Pointwise filter values,
kNN sets, kernel weights, lmax and Chebyshev coefficients rest on the [UPSTREAM] restatement.
"""
from __future__ import annotations

import numbers

import numpy as np
from scipy import sparse
from scipy.sparse.linalg import eigsh
from scipy.spatial.distance import pdist, squareform

__all__ = [
    "knn_kernel",
    "dense_kernel",
    "semantic_kernel_dense",
    "symmetrize",
    "apply_anisotropy",
    "weights_from_kernel",
    "laplacian",
    "estimate_lmax",
    "filter_kernel_fn",
    "cheby_coeff",
    "cheby_op",
    "exact_filter",
    "sample_indicators",
    "build_graph",
    "meld_filter",
    "fit_transform",
    "normalize_densities",
    "OracleGraph",
]


# --------------------------------------------------------------------------------------------
# A2 + A3  kNN search and alpha-decay kernel
# --------------------------------------------------------------------------------------------
def knn_kernel(
    X,
    knn=5,
    decay=40,
    thresh=1e-4,
    search_multiplier=6,
    n_jobs=1,
    algorithm="ball_tree",
    return_intermediates=False,
    distance="euclidean",
    bandwidth=None,
    bandwidth_scale=1.0,
    knn_max=None,
):
    """Directed alpha-decay kernel K (CSR, N x N, includes K_ii = 1).

    ``knn_max`` ([UPSTREAM ``kNNGraph.build_kernel``: ``knn_max = self.knn_max + 1 if self.knn_max else None`` handed to
    ``build_kernel_to_data``, where it caps every ``search_knn`` and ends the re-search with "search out to knn_max"]): a row
    keeps its knn_max nearest cells (besides itself) at most.

    ``bandwidth`` / ``bandwidth_scale`` ([UPSTREAM graphtools ``kNNGraph(bandwidth=, bandwidth_scale=)``, forwarded by reference
    ``meld/meld.py:106,117-118``; ``build_kernel_to_data``: ``if bandwidth is None: bandwidth = distances[:, knn - 1]``, then
    ``bandwidth = bandwidth * bandwidth_scale`` and ``np.maximum(bandwidth, eps)``]): a number or one value per cell replaces
    the adaptive bandwidth; the scale multiplies whichever is used.

    ``distance``: the metric handed to sklearn's ``NearestNeighbors`` ([UPSTREAM graphtools ``kNNGraph.knn_tree``: the ball tree
    with ``metric=self.distance``, ``algorithm="auto"`` when the tree does not take that metric -- "cosine" is brute force]).

    [UPSTREAM graphtools 1.5.x ``kNNGraph.build_kernel`` -> ``build_kernel_to_data(Y=data,
    knn=self.knn + 1)``], reached from reference ``meld/meld.py:273`` (``self.fit``) with the
    kwargs forwarded at ``meld/meld.py:117-118``.

    Steps restated:
      * ``search_knn = min((knn+1) * search_multiplier, N)`` nearest neighbours *including self*
        (sklearn ``NearestNeighbors(algorithm="ball_tree")``, euclidean);
      * ``bandwidth_i = distances[i, knn]`` (k-th non-self neighbour), ``max(., eps)``;
      * ``radius_i = bandwidth_i * (-log thresh)^(1/decay)``;
      * rows whose farthest found neighbour is still inside ``radius_i`` are re-searched with
        6x more neighbours while more than N//10 rows need it, then with a radius search;
      * ``K_ij = exp(-(d_ij / bandwidth_i)^decay)``, NaN -> 1, values < thresh dropped.

    Net semantics (what the GPU path must reproduce): K_ij = exp(-(d_ij/bw_i)^decay) for every j
    for which that value is >= thresh, bw_i = (knn+1)-th smallest distance in row i counting self.
    """
    from sklearn.neighbors import NearestNeighbors

    X = np.ascontiguousarray(X, dtype=np.float64)
    N = X.shape[0]
    if thresh < np.finfo(float).eps:
        thresh = np.finfo(float).eps  # [UPSTREAM kNNGraph.__init__]
    if knn > N - 2:
        knn = N - 2  # [UPSTREAM kNNGraph.__init__] (warns upstream)
    k1 = knn + 1
    if knn_max is not None and knn_max < knn:
        raise ValueError("`knn_max` must be greater than or equal to `knn`")  # [UPSTREAM kNNGraph.__init__]
    knn_max = N if knn_max is None else min(knn_max + 1, N)
    tree = NearestNeighbors(n_neighbors=k1, algorithm=algorithm if distance == "euclidean" else "auto", metric=distance, n_jobs=n_jobs).fit(X)
    if decay is None or thresh == 1:
        # [UPSTREAM graphtools kNNGraph.build_kernel_to_data]: without alpha decay the kernel is the binary
        # connectivity of the knn + 1 nearest neighbours, self included ("unweighted kNN graph")
        K = tree.kneighbors_graph(X, n_neighbors=k1, mode="connectivity").tocsr()
        if return_intermediates:
            dist = tree.kneighbors(X, n_neighbors=k1)[0]
            return K, dict(bandwidth=np.maximum(dist[:, k1 - 1], np.finfo(float).eps), n_updated_first=0)
        return K

    search_knn = min(k1 * search_multiplier, knn_max)
    distances, indices = tree.kneighbors(X, n_neighbors=search_knn)
    if bandwidth is None:
        bandwidth = distances[:, k1 - 1].copy()
    else:
        bandwidth = np.broadcast_to(np.asarray(bandwidth, dtype=np.float64), (N,)).copy()
    bandwidth = bandwidth * bandwidth_scale
    bandwidth = np.maximum(bandwidth, np.finfo(float).eps)
    radius = bandwidth * np.power(-1 * np.log(thresh), 1 / decay)
    update_idx = np.argwhere(np.max(distances, axis=1) < radius).reshape(-1)
    n_updated_first = len(update_idx)

    if len(update_idx) > 0:
        distances = [d for d in distances]
        indices = [i for i in indices]

    search_knn = min(search_knn * search_multiplier, knn_max)
    while len(update_idx) > N // 10 and search_knn < N / 2 and search_knn < knn_max:
        dist_new, ind_new = tree.kneighbors(X[update_idx], n_neighbors=search_knn)
        for i, idx in enumerate(update_idx):
            distances[idx] = dist_new[i]
            indices[idx] = ind_new[i]
        keep = [i for i, d in enumerate(dist_new) if np.max(d) < radius[update_idx[i]]]
        update_idx = update_idx[keep]
        search_knn = min(search_knn * search_multiplier, knn_max)
    if search_knn > N / 2:
        tree = NearestNeighbors(n_neighbors=search_knn, algorithm="brute", metric=distance, n_jobs=n_jobs).fit(X)
    if len(update_idx) > 0:
        if search_knn == knn_max:
            dist_new, ind_new = tree.kneighbors(X[update_idx], n_neighbors=search_knn)
        else:
            dist_new, ind_new = tree.radius_neighbors(X[update_idx, :], radius=np.max(radius[update_idx]))
        for i, idx in enumerate(update_idx):
            distances[idx] = dist_new[i]
            indices[idx] = ind_new[i]

    data = np.concatenate([np.asarray(distances[i]) / bandwidth[i] for i in range(N)])
    cols = np.concatenate([np.asarray(ix) for ix in indices])
    indptr = np.concatenate([[0], np.cumsum([len(d) for d in distances])])
    K = sparse.csr_matrix((data, cols, indptr), shape=(N, N))
    K.data = np.exp(-1 * np.power(K.data, decay))
    K.data = np.where(np.isnan(K.data), 1, K.data)
    K.data[K.data < thresh] = 0
    K = K.tocoo()
    K.eliminate_zeros()
    K = K.tocsr()
    K.sort_indices()
    if return_intermediates:
        return K, dict(bandwidth=bandwidth, radius=radius, n_research_rows=n_updated_first)
    return K


def dense_kernel(X, knn=5, decay=40, thresh=0.0, bandwidth=None, bandwidth_scale=1.0):
    """Dense "exact" kernel used when thresh == 0.

    [UPSTREAM graphtools ``api.Graph`` picks ``TraditionalGraph`` when ``decay is not None and
    thresh == 0``; ``TraditionalGraph.build_kernel``]: pdist/squareform, bandwidth =
    max of the (knn+1) smallest entries per row (self included), ``K = exp(-(pdx/bw)^decay)``,
    NaN -> 1, ``K[K < thresh] = 0``.  Exercised by reference ``test/test_meld.py:59-69``.
    ``bandwidth`` / ``bandwidth_scale`` ([UPSTREAM ``TraditionalGraph.build_kernel``: ``bandwidth = self.bandwidth(pdx)`` for a
    callable -- the one graph class that takes one --, a number or one value per cell otherwise; then
    ``bandwidth = bandwidth * self.bandwidth_scale``], forwarded by reference ``meld/meld.py:106,117-118``).
    """
    X = np.ascontiguousarray(X, dtype=np.float64)
    N = X.shape[0]
    if knn > N - 2:
        knn = N - 2
    pdx = squareform(pdist(X, metric="euclidean"))
    if bandwidth is None:
        knn_dist = np.partition(pdx, knn + 1, axis=1)[:, : knn + 1]
        bandwidth = np.max(knn_dist, axis=1)
    elif callable(bandwidth):
        bandwidth = bandwidth(pdx)
    bandwidth = np.asarray(bandwidth, dtype=np.float64) * bandwidth_scale
    pdx = (pdx.T / bandwidth).T
    K = np.exp(-1 * np.power(pdx, decay))
    K = np.where(np.isnan(K), 1, K)
    K[K < thresh] = 0
    return K


def precomputed_kernel(M, kind, knn=5, decay=40, thresh=1e-4):
    """Kernel from a precomputed N x N matrix [UPSTREAM graphtools ``TraditionalGraph.build_kernel`` with
    ``precomputed="distance" | "affinity" | "adjacency"``; reached through ``GraphEstimator(distance="precomputed_*")``
    from reference ``meld/meld.py:273``]: distances go through the alpha-decay kernel with the bandwidth of
    ``dense_kernel`` (max of the knn+1 smallest entries of a row), an affinity is taken as it is, an adjacency gets a unit
    diagonal; then ``K[K < thresh] = 0``."""
    M = np.asarray(M, dtype=np.float64)
    N = M.shape[0]
    if kind == "distance":
        k = min(knn, N - 2)
        knn_dist = np.partition(M, k + 1, axis=1)[:, : k + 1]
        bandwidth = np.max(knn_dist, axis=1)
        pdx = (M.T / bandwidth).T
        K = np.exp(-1 * np.power(pdx, decay))
        K = np.where(np.isnan(K), 1, K)
    elif kind == "affinity":
        K = M.copy()
    elif kind == "adjacency":
        K = M.copy()
        np.fill_diagonal(K, 1)
    else:
        raise ValueError(kind)
    K[K < thresh] = 0
    return K


def semantic_kernel_dense(X, knn=5, decay=40, thresh=1e-4):
    """Brute-force statement of the kernel's *semantics* (small N only, O(N^2) memory).

    K_ij = v_ij if v_ij >= thresh else 0, v_ij = exp(-(d_ij / bw_i)^decay), bw_i = (knn+1)-th
    smallest distance of row i counting self.  Used by the tests to show that ``knn_kernel``'s
    search/re-search control flow and the GPU candidate/fallback control flow both reduce to it.
    """
    X = np.ascontiguousarray(X, dtype=np.float64)
    N = X.shape[0]
    if knn > N - 2:
        knn = N - 2
    if thresh < np.finfo(float).eps:
        thresh = np.finfo(float).eps
    d = squareform(pdist(X, metric="euclidean"))
    bw = np.maximum(np.sort(d, axis=1)[:, knn], np.finfo(float).eps)
    v = np.exp(-np.power(d / bw[:, None], decay))
    v[v < thresh] = 0
    return sparse.csr_matrix(v), bw


# --------------------------------------------------------------------------------------------
# A4  symmetrise, anisotropy, weights      A5  Laplacian
# --------------------------------------------------------------------------------------------
def symmetrize(K, kernel_symm="+", theta=None):
    """[UPSTREAM graphtools ``BaseGraph.symmetrize_kernel``]: ``K <- (K + K^T) / 2`` with the default kernel_symm="+";
    "*": ``K.multiply(K.T)``; "mnn": ``theta * min(K, K^T) + (1 - theta) * max(K, K^T)`` (theta = 1 when not given)."""
    if kernel_symm == "+":
        return (K + K.T) / 2
    if kernel_symm == "*":
        return K.multiply(K.T) if sparse.issparse(K) else K * K.T
    if kernel_symm == "mnn":
        theta = 1.0 if theta is None else theta
        if sparse.issparse(K):
            return theta * K.minimum(K.T) + (1 - theta) * K.maximum(K.T)
        return theta * np.minimum(K, K.T) + (1 - theta) * np.maximum(K, K.T)
    raise ValueError(kernel_symm)


def apply_anisotropy(K, anisotropy=1):
    """[UPSTREAM graphtools ``BaseGraph.apply_anisotropy``]: ``d = K.sum(1)`` (diagonal
    included); ``K_ij <- K_ij / (d_i d_j)^anisotropy``.  MELD forwards anisotropy=1
    (reference ``meld/meld.py:104,118``)."""
    if anisotropy == 0:
        return K
    if sparse.issparse(K):
        d = np.array(K.sum(1)).flatten()
        K = K.tocoo()
        K.data = K.data / ((d[K.row] * d[K.col]) ** anisotropy)
        return K.tocsr()
    d = K.sum(1)
    return K / (np.outer(d, d) ** anisotropy)


def weights_from_kernel(K):
    """[UPSTREAM graphtools ``PyGSPGraph._build_weight_from_kernel``]: W = K with zero diagonal."""
    if sparse.issparse(K):
        W = K.tolil(copy=True)
        W.setdiag(0)
        W = W.tocsr()
        W.eliminate_zeros()
        W.sort_indices()
        return W
    W = np.array(K, copy=True)
    np.fill_diagonal(W, 0)
    return W


def laplacian(W):
    """[UPSTREAM pygsp 0.5.1 ``Graph.compute_laplacian('combinatorial')``]: dw = W 1, L = D - W.
    (``lap_type`` is stored by the reference but never forwarded -- ``meld/meld.py:113`` -- so the
    Laplacian is always combinatorial.)"""
    if sparse.issparse(W):
        dw = np.ravel(W.sum(1))
        return (sparse.diags(dw, 0) - W).tocsr(), dw
    dw = W.sum(1)
    return np.diag(dw) - W, dw


# --------------------------------------------------------------------------------------------
# A9  lmax
# --------------------------------------------------------------------------------------------
def estimate_lmax(L, dw=None):
    """[UPSTREAM pygsp 0.5.1 ``Graph.estimate_lmax``] (called at reference ``meld/filter.py:39``):
    ``1.01 * eigsh(L, k=1, tol=5e-3, ncv=min(N, 10))``; ``2 max(dw)`` if ARPACK fails."""
    N = L.shape[0]
    try:
        lmax = eigsh(sparse.csr_matrix(L).astype(np.float64), k=1, tol=5e-3, ncv=min(N, 10), return_eigenvectors=False)
        return float(lmax[0]) * 1.01
    except sparse.linalg.ArpackNoConvergence:  # pragma: no cover
        if dw is None:
            dw = np.ravel(abs(L).sum(1)) / 2
        return 2.0 * float(np.max(dw))


# --------------------------------------------------------------------------------------------
# A8  filter kernels     A10  Chebyshev     A10x  exact
# --------------------------------------------------------------------------------------------
def filter_kernel_fn(filter, beta, offset, order, lmax):
    """Spectral kernels of reference ``meld/filter.py:42-53``."""
    if filter.lower() == "laplacian":
        return lambda x: 1 / (1 + (beta * np.abs(x / lmax - offset)) ** order)
    elif filter.lower() == "heat":
        return lambda x: np.exp(-beta * np.abs(x / lmax - offset) ** order)
    raise NotImplementedError


def cheby_coeff(h, lmax, m):
    """[UPSTREAM pygsp 0.5.1 ``filters.approximations.compute_cheby_coeff(f, m)``], N = m + 1
    quadrature points: c_o = 2/N * sum_j h(a1 cos(pi (j+.5)/N) + a2) cos(pi o (j+.5)/N)."""
    N = m + 1
    a1 = (lmax - 0) / 2
    a2 = (lmax + 0) / 2
    c = np.zeros(m + 1)
    tmpN = np.arange(N)
    num = np.cos(np.pi * (tmpN + 0.5) / N)
    for o in range(m + 1):
        c[o] = 2.0 / N * np.dot(h(a1 * num + a2), np.cos(np.pi * o * (tmpN + 0.5) / N))
    return c


def cheby_op(L, lmax, c, signal):
    """[UPSTREAM pygsp 0.5.1 ``filters.approximations.cheby_op``], single filter (Nscales = 1):
    T0 = s; T1 = (L s - a2 s)/a1; r = c0/2 T0 + c1 T1;
    factor = 2/a1 (L - a2 I); Tk = factor T(k-1) - T(k-2); r += ck Tk."""
    c = np.asarray(c)
    M = c.shape[0]
    if M < 2:
        raise TypeError("The coefficients have an invalid shape")
    a1 = float(lmax - 0) / 2.0
    a2 = float(lmax + 0) / 2.0
    signal = np.asarray(signal, dtype=np.float64)
    L = sparse.csr_matrix(L)
    twf_old = signal
    twf_cur = (L.dot(signal) - a2 * signal) / a1
    r = 0.5 * c[0] * twf_old + c[1] * twf_cur
    factor = 2 / a1 * (L - a2 * sparse.eye(L.shape[0]))
    factor = sparse.csr_matrix(factor)
    for k in range(2, M):
        twf_new = factor.dot(twf_cur) - twf_old
        r += c[k] * twf_new
        twf_old = twf_cur
        twf_cur = twf_new
    return r


def exact_filter(L, h_of_lmax, signal):
    """[UPSTREAM pygsp 0.5.1 ``Filter.filter(method='exact')`` + ``compute_fourier_basis``]:
    e, U = eigh(L.toarray()); lmax <- e[-1]; r = U diag(h(e)) U^T s.
    ``h_of_lmax(lmax)`` returns the kernel closure: the reference closure reads ``graph.lmax`` at
    evaluation time (``meld/filter.py:45,50``), i.e. *after* the Fourier basis overwrote it with
    the exact e[-1]."""
    Ld = L.toarray() if sparse.issparse(L) else np.asarray(L)
    e, U = np.linalg.eigh(Ld)
    e[0] = 0 if abs(e[0]) < 1e-10 else e[0]
    lmax = e[-1]
    h = h_of_lmax(lmax)
    s = np.asarray(signal, dtype=np.float64)
    return U @ (h(e)[:, None] * (U.T @ s)), lmax


# --------------------------------------------------------------------------------------------
# A7  indicators     A11 wrap     next#1 normalize_densities
# --------------------------------------------------------------------------------------------
def sample_indicators(sample_labels, sample_normalize=True):
    """Reference ``meld/meld.py:143-191`` + ``:229-232``: columns = sorted unique labels, one-hot,
    each column divided by its sum.  Returns (samples, indicator ndarray [N, p])."""
    labels = np.asarray(getattr(sample_labels, "values", sample_labels))
    if labels.ndim > 1:
        if labels.shape[1] == 1:
            labels = labels.reshape(-1)
        else:
            raise ValueError("sample_labels must be a single column. Got" "shape={}".format(labels.shape))
    samples = np.unique(labels)
    ind = (labels[:, None] == samples[None, :]).astype(np.float64)
    if sample_normalize:
        ind = ind / ind.sum(axis=0)
    return samples, ind


def normalize_densities(sample_densities):
    """Reference ``meld/utils.py:35-47``: L1 row normalisation (sklearn ``normalize(norm='l1')``:
    divide by sum of |.|, rows of zeros left untouched)."""
    a = np.asarray(sample_densities, dtype=np.float64)
    norms = np.abs(a).sum(axis=1)
    norms[norms == 0] = 1.0
    return a / norms[:, None]


# --------------------------------------------------------------------------------------------
# whole path
# --------------------------------------------------------------------------------------------
class OracleGraph:
    """Bag of intermediates (everything the parity tests compare stage by stage)."""

    def __init__(self, K_directed, K, W, L, dw, info=None):
        self.K_directed = K_directed
        self.K = K
        self.W = W
        self.L = L
        self.dw = dw
        self.N = W.shape[0]
        self.info = info or {}
        self.lmax = None


def pca_reduce(X, n_pca, random_state=42, exact=True):
    """[UPSTREAM graphtools/base.py Data._reduce_data] graphtools reduces dense data with
    ``sklearn.decomposition.PCA(n_pca, svd_solver="randomized", random_state=...)`` when
    ``n_pca < min(X.shape)`` and builds the graph on the scores (reference passes ``n_pca`` through at
    ``meld/meld.py:117-118``; default 100).  ``exact=True`` uses the full SVD instead -- the subspace
    the randomized solver approximates -- which is what the product computes; ``exact=False`` is
    the reference's own (approximate, seed-dependent) step."""
    from sklearn.decomposition import PCA

    X = np.asarray(X, dtype=np.float64)
    if n_pca is None or n_pca >= min(X.shape):
        return X
    solver = "full" if exact else "randomized"
    return PCA(n_pca, svd_solver=solver, random_state=random_state).fit_transform(X)


def kernel_to_data(Xq, Yref, knn=5, decay=40, thresh=1e-4, n_jobs=1, algorithm="ball_tree"):
    """Alpha-decay kernel from the rows of ``Xq`` to the rows of ``Yref`` (CSR, len(Xq) x len(Yref)).

    [UPSTREAM graphtools 1.5.x ``kNNGraph.build_kernel_to_data(Y, knn=knn)``] as ``MNNGraph.build_kernel``
    calls it for two different samples: the queries are not among the references, so the bandwidth is the
    distance to the knn-th nearest reference (``distances[:, knn - 1]``; the within-sample kernel uses
    ``knn + 1`` because each point finds itself first), ``radius = bandwidth (-log thresh)^(1/decay)``,
    ``K = exp(-(d / bandwidth)^decay)`` for every reference inside the radius (upstream re-searches rows whose
    neighbour list ends inside the radius; restated here by its net effect, a radius query), values below
    ``thresh`` dropped."""
    from sklearn.neighbors import NearestNeighbors

    Xq = np.ascontiguousarray(Xq, dtype=np.float64)
    Yref = np.ascontiguousarray(Yref, dtype=np.float64)
    if thresh < np.finfo(float).eps:
        thresh = np.finfo(float).eps
    knn = min(knn, Yref.shape[0])  # [UPSTREAM]: warns and clips knn to the size of the reference sample
    tree = NearestNeighbors(n_neighbors=knn, algorithm=algorithm, metric="euclidean", n_jobs=n_jobs).fit(Yref)
    if decay is None or thresh == 1:
        return tree.kneighbors_graph(Xq, n_neighbors=knn, mode="connectivity").tocsr()
    dist = tree.kneighbors(Xq, n_neighbors=knn)[0]
    bandwidth = np.maximum(dist[:, knn - 1], np.finfo(float).eps)
    radius = bandwidth * np.power(-1 * np.log(thresh), 1 / decay)
    rows, cols, vals = [], [], []
    for lo in range(0, Xq.shape[0], 4096):
        hi = min(Xq.shape[0], lo + 4096)
        dd, ii = tree.radius_neighbors(Xq[lo:hi], radius=float(np.max(radius[lo:hi])) * (1 + 1e-12))
        for r in range(hi - lo):
            v = np.exp(-1 * np.power(dd[r] / bandwidth[lo + r], decay))
            v = np.where(np.isnan(v), 1, v)
            keep = v >= thresh
            rows.append(np.full(int(keep.sum()), lo + r, dtype=np.int64))
            cols.append(ii[r][keep].astype(np.int64))
            vals.append(v[keep])
    K = sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(Xq.shape[0], Yref.shape[0]))
    K.sort_indices()
    return K


def mnn_kernel(X, sample_idx, knn=5, decay=40, thresh=1e-4, beta=1.0, n_jobs=1, algorithm="ball_tree"):
    """Directed mutual-nearest-neighbours kernel of ``graphtools.Graph(data, sample_idx=...)`` (CSR, N x N).

    [UPSTREAM graphtools 1.5.x ``MNNGraph.build_kernel``], reached from reference ``meld/meld.py:117-118``
    when the caller forwards ``sample_idx`` (``test/test_meld.py:34``, ``test/test_utils.py:11``):
      * one kNN graph per sample (``kernel_symm="+"``, no anisotropy): the diagonal block of sample i is its
        symmetrised alpha-decay kernel ``(k_i + k_i^T) / 2`` (diagonal 1);
      * the block from sample i to sample j != i is ``kernel_to_data(X_i, X_j, knn)`` with every row scaled by
        ``min(1, within_i / between_ij) * beta`` -- row sum of the diagonal block over row sum of this block;
    the assembled matrix then goes through the common ``symmetrize`` / ``apply_anisotropy`` steps.
    PARITY: unpinned (the reference only checks that this path runs)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    sample_idx = np.asarray(sample_idx)
    if sample_idx.shape[0] != X.shape[0]:
        raise ValueError("sample_idx ({}) must be the same length as data ({})".format(sample_idx.shape[0], X.shape[0]))
    samples = np.unique(sample_idx)
    if len(samples) == 1:
        raise ValueError("sample_idx must contain more than one unique value")
    N = X.shape[0]
    members = [np.nonzero(sample_idx == s)[0] for s in samples]
    K = sparse.lil_matrix((N, N))
    blocks = []
    for i, mi in enumerate(members):
        Kii = symmetrize(knn_kernel(X[mi], knn=knn, decay=decay, thresh=thresh, n_jobs=n_jobs, algorithm=algorithm)).tocsr()
        within = np.asarray(Kii.sum(1)).ravel()
        blocks.append((mi, mi, Kii.tocoo()))
        for j, mj in enumerate(members):
            if i == j:
                continue
            Kij = kernel_to_data(X[mi], X[mj], knn=knn, decay=decay, thresh=thresh, n_jobs=n_jobs, algorithm=algorithm)
            between = np.asarray(Kij.sum(1)).ravel()
            scale = np.minimum(1, within / between) * beta
            blocks.append((mi, mj, Kij.multiply(scale[:, None]).tocoo()))
    rows = np.concatenate([mi[b.row] for mi, mj, b in blocks])
    cols = np.concatenate([mj[b.col] for mi, mj, b in blocks])
    vals = np.concatenate([b.data for mi, mj, b in blocks])
    K = sparse.csr_matrix((vals, (rows, cols)), shape=(N, N))
    K.sort_indices()
    return K


def mnn_kernel_dense(X, sample_idx, knn=5, decay=40, beta=1.0):
    """The MNN kernel with ``thresh == 0`` (dense N x N).

    [UPSTREAM graphtools 1.5.x ``MNNGraph.build_kernel``]: with ``thresh = 0`` and a decay every per-sample subgraph is the
    "exact" ``TraditionalGraph`` (``dense_kernel``, symmetrised with '+'), and the block from sample i to sample j is
    ``TraditionalGraph.build_kernel_to_data(X_i, knn)`` of subgraph j: ``cdist``, bandwidth = the largest of the ``knn``
    smallest entries of a row (no self among the references), ``exp(-(d / bw)^decay)``, NaN -> 1, nothing dropped; rows
    scaled by ``min(1, within_i / between_ij) * beta`` as in ``mnn_kernel``.  PARITY: unpinned."""
    from scipy.spatial.distance import cdist

    X = np.ascontiguousarray(X, dtype=np.float64)
    sample_idx = np.asarray(sample_idx)
    if sample_idx.shape[0] != X.shape[0]:
        raise ValueError("sample_idx ({}) must be the same length as data ({})".format(sample_idx.shape[0], X.shape[0]))
    samples = np.unique(sample_idx)
    if len(samples) == 1:
        raise ValueError("sample_idx must contain more than one unique value")
    N = X.shape[0]
    members = [np.nonzero(sample_idx == s)[0] for s in samples]
    K = np.zeros((N, N))
    for i, mi in enumerate(members):
        Kii = symmetrize(dense_kernel(X[mi], knn=knn, decay=decay, thresh=0.0))
        K[np.ix_(mi, mi)] = Kii
        within = Kii.sum(1)
        for j, mj in enumerate(members):
            if i == j:
                continue
            pdx = cdist(X[mi], X[mj], metric="euclidean")
            kk = min(knn, len(mj) - 1)  # (np.partition needs knn < len(row); upstream fails beyond that)
            bandwidth = np.max(np.partition(pdx, kk, axis=1)[:, :kk], axis=1)
            Kij = np.exp(-1 * np.power((pdx.T / bandwidth).T, decay))
            Kij = np.where(np.isnan(Kij), 1, Kij)
            scale = np.minimum(1, within / Kij.sum(1)) * beta
            K[np.ix_(mi, mj)] = Kij * scale[:, None]
    return K


def build_graph(X, knn=5, decay=40, thresh=1e-4, anisotropy=1, n_jobs=1, algorithm="ball_tree", n_pca=None, sample_idx=None, distance="euclidean",
                bandwidth=None, bandwidth_scale=1.0, knn_max=None, kernel_symm="+", theta=None):
    """A1-A5: data -> OracleGraph.  ``n_pca`` (None = off; graphtools only reduces when
    ``n_pca < min(X.shape)`` [UPSTREAM], which none of the BASELINE configs trigger) runs
    ``pca_reduce`` first.  ``sample_idx``: the MNN kernel between samples (``mnn_kernel``)."""
    if n_pca is not None:
        X = pca_reduce(X, n_pca)
    if distance != "euclidean" and (sample_idx is not None or thresh == 0):
        raise NotImplementedError("the oracle restates non-euclidean distances for the kNN graph only")
    if sample_idx is not None and thresh == 0 and decay is not None:
        Kd = mnn_kernel_dense(X, sample_idx, knn=knn, decay=decay)
        K = apply_anisotropy(symmetrize(Kd), anisotropy)
        W = weights_from_kernel(K)
        L, dw = laplacian(W)
        return OracleGraph(Kd, K, W, L, dw)
    if sample_idx is not None:
        Kd = mnn_kernel(X, sample_idx, knn=knn, decay=decay, thresh=thresh, n_jobs=n_jobs, algorithm=algorithm)
        K = apply_anisotropy(symmetrize(Kd), anisotropy).tocsr()
        K.sort_indices()
        W = weights_from_kernel(K)
        L, dw = laplacian(W)
        return OracleGraph(Kd, K, W, L, dw)
    if thresh == 0 and decay is not None:  # ([UPSTREAM graphtools api.Graph]: decay=None picks the kNN graph before thresh is looked at)
        Kd = dense_kernel(X, knn=knn, decay=decay, thresh=0.0, bandwidth=bandwidth, bandwidth_scale=bandwidth_scale)
        K = apply_anisotropy(symmetrize(Kd, kernel_symm, theta), anisotropy)
        W = weights_from_kernel(K)
        L, dw = laplacian(W)
        return OracleGraph(Kd, K, W, L, dw)
    Kd, info = knn_kernel(X, knn=knn, decay=decay, thresh=thresh, n_jobs=n_jobs, algorithm=algorithm, return_intermediates=True, distance=distance,
                          bandwidth=bandwidth, bandwidth_scale=bandwidth_scale, knn_max=knn_max)
    K = apply_anisotropy(symmetrize(Kd, kernel_symm, theta).tocsr(), anisotropy).tocsr()
    K.eliminate_zeros()
    K.sort_indices()
    W = weights_from_kernel(K)
    L, dw = laplacian(W)
    return OracleGraph(Kd, K, W, L, dw, info)


def meld_filter(signal, graph, filter="heat", beta=60, offset=0, order=1, solver="chebyshev", chebyshev_order=50, lmax=None):
    """Reference ``meld/filter.py:5-61``.  ``lmax=`` injects a precomputed value (pygsp's
    ``estimate_lmax`` is a no-op when ``_lmax`` is already set [UPSTREAM])."""
    if lmax is None:
        lmax = graph.lmax if graph.lmax is not None else estimate_lmax(graph.L, graph.dw)
    graph.lmax = lmax
    if filter.lower() not in ("heat", "laplacian"):
        raise NotImplementedError
    if solver == "exact":
        r, lmax_exact = exact_filter(graph.L, lambda lm: filter_kernel_fn(filter, beta, offset, order, lm), signal)
        graph.lmax = lmax_exact
        return r
    h = filter_kernel_fn(filter, beta, offset, order, lmax)
    c = cheby_coeff(h, lmax, chebyshev_order)
    return cheby_op(graph.L, lmax, c, signal)


def fit_transform(
    X,
    sample_labels,
    beta=60,
    offset=0,
    order=1,
    filter="heat",
    solver="chebyshev",
    chebyshev_order=50,
    sample_normalize=True,
    anisotropy=1,
    knn=5,
    decay=40,
    thresh=1e-4,
    n_jobs=1,
    algorithm="ball_tree",
    lmax=None,
    return_graph=False,
    distance="euclidean",
):
    """``meld.MELD(**params).fit_transform(X, sample_labels)`` -- reference
    ``meld/meld.py:252-274`` with the constructor defaults of ``meld/meld.py:94-107`` and the
    graphtools defaults knn=5, decay=40, thresh=1e-4 [UPSTREAM GraphEstimator]."""
    G = build_graph(X, knn=knn, decay=decay, thresh=thresh, anisotropy=anisotropy, n_jobs=n_jobs, algorithm=algorithm, distance=distance)
    samples, ind = sample_indicators(sample_labels, sample_normalize)
    dens = meld_filter(ind, G, filter=filter, beta=beta, offset=offset, order=order, solver=solver, chebyshev_order=chebyshev_order, lmax=lmax)
    if return_graph:
        return samples, dens, G
    return samples, dens


# --------------------------------------------------------------------------------------------
# synthetic inputs shared by tests and bench
# --------------------------------------------------------------------------------------------
def synthetic_cells(n_cells, n_dims=50, seed=0, latent_dim=10, n_clusters=20):
    """Seeded low-intrinsic-dimension mixture of SURVEY.md section 8(d): 20 cluster centres
    ~ N(0, 4 I_10), points = centre + N(0, I_10), embedded in ``n_dims`` by a fixed random
    orthonormal map, plus N(0, 0.05^2) isotropic noise.  Labels: Bernoulli(expit(latent_0)) as in
    reference ``meld/benchmark.py:174,181-184`` / ``test/test_meld.py:56-57``."""
    rng = np.random.default_rng(seed)
    latent_dim = min(latent_dim, n_dims)
    centres = rng.normal(0.0, 2.0, size=(n_clusters, latent_dim))
    assign = rng.integers(0, n_clusters, size=n_cells)
    latent = centres[assign] + rng.normal(0.0, 1.0, size=(n_cells, latent_dim))
    q, _ = np.linalg.qr(rng.normal(size=(n_dims, latent_dim)))
    X = latent @ q.T + rng.normal(0.0, 0.05, size=(n_cells, n_dims))
    p = 1.0 / (1.0 + np.exp(-latent[:, 0]))
    labels = np.where(rng.random(n_cells) < p, "expt", "ctrl")
    return np.ascontiguousarray(X, dtype=np.float64), labels


# ---------------------------------------------------------------------------------------------
# VertexFrequencyCluster at small N -- restatement of the reference's
# ``meld/cluster.py`` (dense windowed graph Fourier transform + PCA + KMeans).  Test infrastructure,
# like the rest of this file.
# ---------------------------------------------------------------------------------------------
def diff_op(K):
    """[UPSTREAM graphtools ``BaseGraph.diff_op``]: the kernel (diagonal included), l1-normalised by
    rows -- the base window of reference ``meld/cluster.py:213-215``."""
    K = np.asarray(K.todense()) if sparse.issparse(K) else np.asarray(K, dtype=np.float64)
    return K / K.sum(axis=1, keepdims=True)


def fourier_basis(L):
    """[UPSTREAM pygsp ``Graph.compute_fourier_basis``]: full eigendecomposition of the (combinatorial)
    Laplacian, eigenvalues ascending (reference ``meld/cluster.py:235-236``)."""
    Ld = np.asarray(L.todense()) if sparse.issparse(L) else np.asarray(L, dtype=np.float64)
    e, U = np.linalg.eigh(Ld)
    return e, U


def _l2_normalize_columns(M):
    """sklearn ``preprocessing.normalize(M, "l2", axis=0)``: unit-norm columns, zero columns kept."""
    nrm = np.sqrt((M * M).sum(axis=0, keepdims=True))
    nrm[nrm == 0] = 1.0
    return M / nrm


def vfc_windows(P, window_sizes):
    """Reference ``meld/cluster.py:179-194`` (dyadic sizes: repeated squaring) and ``:158-177``
    (arbitrary sizes: matrix power); every window = l2-normalised columns of P^t, transposed."""
    window_sizes = np.asarray(window_sizes)
    if np.all(np.diff(np.log2(window_sizes)) == 1):
        out, cur = [], P
        out.append(_l2_normalize_columns(cur).T)
        for _ in range(len(window_sizes) - 1):
            cur = cur @ cur
            out.append(_l2_normalize_columns(cur).T)
        return out
    return [_l2_normalize_columns(np.linalg.matrix_power(P, int(t))).T for t in window_sizes]


def vfc_spectrogram(U, windows, s):
    """Reference ``meld/cluster.py:98-156``: sum over the windows of tanh|normalize(U^T (W_t * s), axis=0)^T|."""
    out = np.zeros((windows[0].shape[1], U.shape[1]))
    for W in windows:
        C = _l2_normalize_columns(U.T @ (W * s[None, :]))
        out += np.tanh(np.abs(C.T))
    return out


def vfc_transform(K, L, sample_indicator, likelihood=None, window_sizes=None, center=True, likelihood_bias=1):
    """Reference ``meld/cluster.py:207-309`` (fit + transform).  Returns (spectrogram, combined or None)."""
    if window_sizes is None:
        window_sizes = np.power(2, np.arange(9))
    windows = vfc_windows(diff_op(K), window_sizes)
    _, U = fourier_basis(L)
    s = np.array(sample_indicator, dtype=np.float64)
    if center:
        s = s - s.mean()
    if s.ndim == 1:
        spec = vfc_spectrogram(U, windows, s)
    else:
        spec = np.hstack([vfc_spectrogram(U, windows, s[:, i]) for i in range(s.shape[1])])
    combined = None
    if likelihood is not None:
        lik = np.array(likelihood, dtype=np.float64)
        spec_n = spec / np.linalg.norm(spec)
        ees_n = lik / np.linalg.norm(lik, ord=2, axis=0) * likelihood_bias
        combined = np.c_[spec_n, ees_n]
    return spec, combined


def sort_clusters_by_values(clusters, values):
    """[UPSTREAM scprep ``utils.sort_clusters_by_values``]: relabel clusters 0..k-1 by ascending mean of
    ``values`` over their members (reference ``meld/cluster.py:346-353``)."""
    clusters = np.asarray(clusters)
    values = np.asarray(values, dtype=np.float64)
    uniq = np.unique(clusters)
    means = np.array([np.mean(values[clusters == c]) for c in uniq])
    remap = {c: i for i, c in enumerate(uniq[np.argsort(means)])}
    return np.array([remap[c] for c in clusters])


def vfc_predict(data, n_clusters, values, random_state=None):
    """Reference ``meld/cluster.py:315-357``: PCA(n_clusters) -> KMeans(n_clusters, n_init=10) -> clusters
    sorted by their mean likelihood / indicator."""
    from sklearn.cluster import KMeans
    from sklearn.decomposition import PCA

    Y = PCA(n_clusters).fit_transform(data)
    labels = KMeans(n_clusters=n_clusters, n_init=10, random_state=random_state).fit_predict(Y)
    return sort_clusters_by_values(labels, values)
