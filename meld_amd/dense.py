"""Small-N dense side paths of the reference, on the GPU: the ``thresh=0`` "exact" graph and
``solver="exact"``.

These are O(N^2) memory / O(N^3) time in the reference as well (it uses them at N = 1000 in its
only known-answer test, ``test/test_meld.py:43-81``); they are not the accelerated hot path.  They
run on dense PyTorch-ROCm linear algebra (rocBLAS / rocSOLVER ``eigh``) rather than hand-written
kernels, and exist so that the reference's known-answer test can be replayed through the product.

* ``build_dense_graph``: [UPSTREAM graphtools ``TraditionalGraph.build_kernel``] -- pairwise
  distances, bandwidth = (knn+1)-th smallest per row (self counted), K = exp(-(d/bw)^decay),
  NaN -> 1, no threshold; then the same symmetrise / anisotropy / zero-diagonal steps.
* ``exact_filter``: [UPSTREAM pygsp ``Filter.filter(method='exact')``] -- e, U = eigh(L);
  lmax <- e[-1] (pygsp overwrites the Lanczos estimate, and the reference's kernel closure reads
  ``graph.lmax`` afterwards, ``meld/filter.py:45,50``); r = U diag(h(e)) U^T s.
"""
from __future__ import annotations

import numpy as np
import torch

from .graph import DeviceGraph

__all__ = ["build_dense_graph", "build_dense_mnn_graph", "build_dense_knn_graph", "build_precomputed_graph", "exact_filter", "DENSE_MAX_N"]

DENSE_MAX_N = 16384


def _graph_from_dense_kernel(K, anisotropy, bw, info, symm=(0, 0.0)):
    """Directed dense kernel K [N, N] (device fp64) -> DeviceGraph: symmetrisation ((K + K^T) / 2 by default; ``symm`` as
    ``graph.symm_code`` returns it), anisotropy, zero diagonal
    [UPSTREAM graphtools ``BaseGraph._build_kernel`` / ``symmetrize_kernel`` / ``apply_anisotropy``]."""
    N = int(K.shape[0])
    if symm[0] == 1:
        K = K * K.T
    elif symm[0] == 2:
        K = symm[1] * torch.minimum(K, K.T) + (1.0 - symm[1]) * torch.maximum(K, K.T)
    else:
        K = (K + K.T) / 2
    if anisotropy != 0:
        dsum = K.sum(1)
        K = K / torch.pow(dsum[:, None] * dsum[None, :], anisotropy)
        ksum = dsum
    else:
        ksum = K.sum(1)
    W = K.clone()
    W.fill_diagonal_(0.0)
    nz = W != 0
    counts = nz.sum(1)
    rowptr = torch.zeros(N + 1, dtype=torch.int64, device=K.device)
    rowptr[1:] = torch.cumsum(counts, 0)
    idx = torch.nonzero(nz)  # row-major order = CSR order with sorted columns
    col = idx[:, 1].to(torch.int32).contiguous()
    val = W[nz].contiguous()
    dw = W.sum(1).contiguous()
    G = DeviceGraph(rowptr, col, val, dw, ksum=ksum, anisotropy=anisotropy, info=dict(info, N=N, dense=True, nnz=int(col.shape[0]), n_flagged_rows=0))
    G.bandwidth = bw
    G._kdiag = torch.diagonal(K).clone()
    return G


def _alpha_decay_dense(D, knn, decay, thresh):
    """[UPSTREAM graphtools ``TraditionalGraph.build_kernel``]: bandwidth = the (knn+1)-th smallest entry of a row (self
    counted), K = exp(-(d / bw)^decay), NaN -> 1, K < thresh -> 0."""
    bw = torch.kthvalue(D, knn + 1, dim=1).values
    if decay is None or decay == float("inf"):  # the unweighted kNN graph: connectivity of the knn + 1 nearest cells
        return (D <= bw[:, None]).to(torch.float64), bw
    K = torch.exp(-torch.pow(D / bw[:, None], decay))
    K = torch.where(torch.isnan(K), torch.ones_like(K), K)
    if thresh > 0:
        K = torch.where(K < thresh, torch.zeros_like(K), K)
    return K, bw


def build_dense_graph(X, knn=5, decay=40, anisotropy=1, symm=(0, 0.0), bandwidth=None, bandwidth_scale=1.0):
    """``bandwidth`` ([UPSTREAM ``TraditionalGraph.build_kernel``]: a number, one value per cell, or a CALLABLE that is handed the
    N x N matrix of pairwise distances (a host array, as upstream hands it a numpy array) and returns either) replaces the distance
    to the knn-th neighbour; ``bandwidth_scale`` multiplies whichever is used."""
    N = int(X.shape[0])
    if N > DENSE_MAX_N:
        raise ValueError("thresh=0 builds a dense {0}x{0} graph; the limit is N <= {1}".format(N, DENSE_MAX_N))
    if knn > N - 2:
        knn = N - 2
    X = X.to(torch.float64)
    D = torch.cdist(X, X, p=2.0, compute_mode="donot_use_mm_for_euclid_dist")
    D.fill_diagonal_(0.0)
    if bandwidth is None and float(bandwidth_scale) == 1.0:
        K, bw = _alpha_decay_dense(D, knn, decay, 0.0)
    else:
        if bandwidth is None:
            bw = torch.kthvalue(D, knn + 1, dim=1).values
        else:
            b = bandwidth(D.cpu().numpy()) if callable(bandwidth) else bandwidth
            bw = torch.as_tensor(np.asarray(b, dtype=np.float64)).to(D.device)
            if bw.dim() == 0:
                bw = bw.expand(N)
            if tuple(bw.shape) != (N,):
                raise ValueError("bandwidth must be a number or have one entry per cell ({}), got shape {}".format(N, tuple(bw.shape)))
        bw = bw * float(bandwidth_scale)
        K = torch.exp(-torch.pow(D / bw[:, None], decay))
        K = torch.where(torch.isnan(K), torch.ones_like(K), K)
    return _graph_from_dense_kernel(K, anisotropy, bw, dict(knn=int(knn)), symm=symm)


def build_dense_mnn_graph(X, sample_idx, knn=5, decay=40, anisotropy=1, beta=1.0):
    """``sample_idx`` with ``thresh=0``: graphtools' MNN kernel over "exact" per-sample subgraphs
    [UPSTREAM ``MNNGraph.build_kernel`` -> ``TraditionalGraph.build_kernel`` / ``build_kernel_to_data``] -- the block of a
    sample with itself is its symmetrised dense alpha-decay kernel, the block from sample i to sample j uses the distance
    to the knn-th cell of j as bandwidth and has its rows scaled by min(1, within_i / between_ij) * beta; nothing dropped."""
    N = int(X.shape[0])
    if N > DENSE_MAX_N:
        raise ValueError("thresh=0 builds a dense {0}x{0} graph; the limit is N <= {1}".format(N, DENSE_MAX_N))
    sample_idx = np.asarray(sample_idx)
    if sample_idx.ndim != 1 or sample_idx.shape[0] != N:
        raise ValueError("sample_idx ({}) must be the same length as data ({})".format(sample_idx.shape[0], N))
    samples, codes = np.unique(sample_idx, return_inverse=True)
    if len(samples) == 1:
        raise ValueError("sample_idx must contain more than one unique value")
    X = X.to(torch.float64)
    members = [torch.from_numpy(np.nonzero(codes == s)[0]).to(X.device) for s in range(len(samples))]
    for s, m in zip(samples, members):
        if m.shape[0] < 3:
            raise ValueError("sample {!r} has {} cells; every sample needs at least 3".format(s, int(m.shape[0])))
    parts = [X.index_select(0, m) for m in members]
    K = torch.zeros(N, N, dtype=torch.float64, device=X.device)
    for i, (mi, Xi) in enumerate(zip(members, parts)):
        D = torch.cdist(Xi, Xi, p=2.0, compute_mode="donot_use_mm_for_euclid_dist")
        D.fill_diagonal_(0.0)
        Kii, _ = _alpha_decay_dense(D, min(int(knn), int(Xi.shape[0]) - 2), decay, 0.0)
        Kii = (Kii + Kii.T) / 2
        K[mi[:, None], mi[None, :]] = Kii
        within = Kii.sum(1)
        for j, (mj, Xj) in enumerate(zip(members, parts)):
            if i == j:
                continue
            D = torch.cdist(Xi, Xj, p=2.0, compute_mode="donot_use_mm_for_euclid_dist")
            bw = torch.kthvalue(D, min(int(knn), int(Xj.shape[0]) - 1), dim=1).values
            Kij = torch.exp(-torch.pow(D / bw[:, None], decay))
            Kij = torch.where(torch.isnan(Kij), torch.ones_like(Kij), Kij)
            scale = torch.clamp(within / Kij.sum(1), max=1.0) * float(beta)
            K[mi[:, None], mj[None, :]] = Kij * scale[:, None]
    return _graph_from_dense_kernel(K, anisotropy, None, dict(knn=int(knn), graph="mnn", n_samples=len(samples)))


# metrics that do not reduce to the euclidean search: pairwise distances by the library (torch.cdist), then the same kernel
_CDIST_P = {"manhattan": 1.0, "cityblock": 1.0, "l1": 1.0, "chebyshev": float("inf")}


def build_dense_knn_graph(X, knn, decay, thresh, anisotropy=1, symm=(0, 0.0), metric="euclidean"):
    """The SPARSE kernel's semantics -- K_ij = exp(-(d_ij / bw_i)^decay) wherever that is >= thresh, bw_i = distance to the
    knn-th neighbour [UPSTREAM kNNGraph.build_kernel_to_data] -- evaluated densely: the route for ``knn`` beyond the 126 the
    candidate lists of the search kernel hold (graphtools has no such limit), up to ``DENSE_MAX_N`` cells."""
    N = int(X.shape[0])
    if N > DENSE_MAX_N:
        raise NotImplementedError("knn={} beyond the 128 entries the search kernel's candidate lists hold, or a metric that does not "
                                  "reduce to the euclidean search ({}): the dense route that serves such graphs is limited to "
                                  "N <= {}".format(knn, metric, DENSE_MAX_N))
    knn = min(int(knn), N - 2)
    X = X.to(torch.float64)
    if metric in ("euclidean", "l2"):
        D = torch.cdist(X, X, p=2.0, compute_mode="donot_use_mm_for_euclid_dist")
    else:
        D = torch.cdist(X, X, p=_CDIST_P[metric])
    D.fill_diagonal_(0.0)
    K, bw = _alpha_decay_dense(D, knn, decay, max(float(thresh), float(np.finfo(float).eps)))
    return _graph_from_dense_kernel(K, anisotropy, bw, dict(knn=int(knn), dense_knn=True, metric=metric), symm=symm)


def build_precomputed_graph(M, kind, knn=5, decay=40, thresh=1e-4, anisotropy=1, symm=(0, 0.0)):
    """A graph from a precomputed N x N matrix (``MELD(distance="precomputed" | "precomputed_distance" |
    "precomputed_affinity").fit(M)``: [UPSTREAM graphtools ``GraphEstimator._parse_input`` -> ``Graph(precomputed=...)`` ->
    ``TraditionalGraph.build_kernel``], reached from reference ``meld/meld.py:273``).  ``kind``: "distance" (pairwise
    distances: the alpha-decay kernel with the (knn+1)-th smallest entry of a row as bandwidth, thresholded), "affinity"
    (the kernel itself) or "adjacency" (the kernel without its diagonal: set to 1).  Dense, like the reference's."""
    if M.dim() != 2 or M.shape[0] != M.shape[1]:
        raise ValueError("Precomputed {} must be a square matrix. {} was given".format(kind, tuple(M.shape)))
    N = int(M.shape[0])
    if N > DENSE_MAX_N:
        raise ValueError("a precomputed matrix is handled densely; the limit is N <= {}".format(DENSE_MAX_N))
    M = M.to(torch.float64)
    if bool((M < 0).any()):
        raise ValueError("Precomputed {} should be non-negative".format(kind))
    bw = None
    if kind == "distance":
        k = min(int(knn), N - 2)
        K, bw = _alpha_decay_dense(M, k, float("inf") if decay is None else decay, thresh)
    elif kind == "affinity":
        K = torch.where(M < thresh, torch.zeros_like(M), M) if thresh > 0 else M.clone()
    elif kind == "adjacency":
        K = M.clone()
        K.fill_diagonal_(1.0)
        if thresh > 0:
            K = torch.where(K < thresh, torch.zeros_like(K), K)
    else:
        raise ValueError("Precomputed value {} not recognized. Choose from ['distance', 'affinity', 'adjacency']".format(kind))
    return _graph_from_dense_kernel(K, anisotropy, bw, dict(knn=int(knn), precomputed=kind), symm=symm)


def exact_filter(graph, sig, kernel_of_lmax):
    """sig: ndarray [N, p].  Returns ndarray [N, p].  Sets ``graph.lmax`` to the exact e[-1]."""
    if graph.n_rows != graph.N:
        raise NotImplementedError("solver='exact' is not available on a sharded graph")
    N = graph.N
    if N > DENSE_MAX_N:
        raise ValueError("solver='exact' needs a dense eigendecomposition; the limit is N <= {}".format(DENSE_MAX_N))
    dev = graph.val.device
    rows = torch.repeat_interleave(torch.arange(N, device=dev), graph.rowptr[1 : N + 1] - graph.rowptr[:N])
    W = torch.zeros(N, N, dtype=torch.float64, device=dev)
    W[rows, graph.col.to(torch.int64)] = graph.val
    L = torch.diag(graph.dw_dev) - W
    e, U = torch.linalg.eigh(L)
    e = e.clone()
    if abs(float(e[0])) < 1e-10:
        e[0] = 0.0
    lmax = float(e[-1])
    graph.lmax = lmax
    h = kernel_of_lmax(lmax)
    he = torch.from_numpy(np.asarray(h(e.cpu().numpy()), dtype=np.float64)).to(dev)
    s = torch.from_numpy(np.ascontiguousarray(sig, dtype=np.float64)).to(dev)
    perm = getattr(graph, "perm", None)
    if perm is not None:
        s = s.index_select(0, perm)
    r = U @ (he[:, None] * (U.T @ s))
    if perm is not None:
        out = torch.empty_like(r)
        out[perm] = r
        r = out
    return r.cpu().numpy()
