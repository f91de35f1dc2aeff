"""PCA front-end of the graph builder (reference ``meld/meld.py:117-118``
passes ``n_pca`` to graphtools, whose ``Data._reduce_data`` [UPSTREAM graphtools/base.py] fits
``sklearn.decomposition.PCA(n_pca, svd_solver="randomized")`` when ``n_pca < min(X.shape)`` and
builds the graph on the projected data).

Device implementation on dense PyTorch-ROCm linear algebra (rocBLAS GEMMs + rocSOLVER ``eigh``; plain
library calls, no hand-written kernel -- the projection is GEMM-shaped and runs once per fit):

* G <= EXACT_MAX features: the centred covariance  C = Xc^T Xc  is accumulated in row chunks (one
  fp64 GEMM each, no second copy of X), ``eigh(C)`` gives the exact top-k subspace, and the scores
  are  Y = Xc V  (another chunked GEMM).  Exact and deterministic: the reference's randomized solver
  approximates this subspace to its own tolerance, so parity is defined against the exact PCA
  (oracle: sklearn ``svd_solver="full"``); distances -- all the graph sees -- do not depend on
  the sign or rotation conventions of the components.
* N <= EXACT_MAX samples (wide data): the same through the Gram matrix  K = Xc Xc^T.
* otherwise: randomized range finder (Halko et al.: Gaussian sketch of k + 10 columns, 4 power
  iterations with QR re-orthonormalisation, small SVD), seeded.
"""
from __future__ import annotations

import torch

__all__ = ["pca_project", "EXACT_MAX"]

EXACT_MAX = 8192
_CHUNK_BYTES = 1 << 30  # rows per GEMM chunk are sized to ~1 GiB of fp64


def _row_chunks(n_rows, n_cols):
    step = max(1, _CHUNK_BYTES // (8 * max(n_cols, 1)))
    for lo in range(0, n_rows, step):
        yield lo, min(n_rows, lo + step)


def _flip_signs(V):
    """Component sign convention (sklearn >= 1.5, ``svd_flip(u_based_decision=False)``): the loading
    of largest magnitude in every component is positive."""
    idx = torch.argmax(V.abs(), dim=0)
    sgn = torch.sign(V[idx, torch.arange(V.shape[1], device=V.device)])
    sgn[sgn == 0] = 1.0
    return V * sgn


def pca_project(X, n_components, seed=42, return_model=False):
    """X: fp64 device tensor [N, G].  Returns the scores Y [N, n_components] (fp64, same device) of the
    top principal components of the column-centred data (and, optionally, (mean, components [G, k]))."""
    if X.dim() != 2:
        raise ValueError("Expected a 2D data matrix, got shape {}".format(tuple(X.shape)))
    N, G = int(X.shape[0]), int(X.shape[1])
    k = int(n_components)
    if not 1 <= k <= min(N, G):
        raise ValueError("n_components={} must lie in [1, min(N, G)={}]".format(k, min(N, G)))
    X = X.to(torch.float64)
    dev = X.device
    mean = X.mean(dim=0)

    if G <= EXACT_MAX:
        C = torch.zeros(G, G, dtype=torch.float64, device=dev)
        for lo, hi in _row_chunks(N, G):
            Xc = X[lo:hi] - mean
            C.addmm_(Xc.T, Xc)
        _, evec = torch.linalg.eigh(C)  # ascending
        V = _flip_signs(evec[:, -k:].flip(1).contiguous())  # [G, k], descending variance
    elif N <= EXACT_MAX:
        Xc = X - mean
        K = Xc @ Xc.T
        ev, evec = torch.linalg.eigh(K)
        U = evec[:, -k:].flip(1)
        s = torch.sqrt(torch.clamp(ev[-k:].flip(0), min=0.0))
        V = Xc.T @ (U / torch.where(s > 0, s, torch.ones_like(s)))
        V = _flip_signs(V.contiguous())
    else:
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(seed))
        r = min(k + 10, min(N, G))
        Q = torch.randn(G, r, dtype=torch.float64, device=dev, generator=gen)

        def times_xc(M):  # Xc @ M, chunked
            out = torch.empty(N, M.shape[1], dtype=torch.float64, device=dev)
            for lo, hi in _row_chunks(N, G):
                out[lo:hi] = (X[lo:hi] - mean) @ M
            return out

        def times_xct(M):  # Xc^T @ M, chunked
            out = torch.zeros(G, M.shape[1], dtype=torch.float64, device=dev)
            for lo, hi in _row_chunks(N, G):
                out.addmm_((X[lo:hi] - mean).T, M[lo:hi])
            return out

        Y = times_xc(Q)
        for _ in range(4):
            Y, _ = torch.linalg.qr(Y)
            Z, _ = torch.linalg.qr(times_xct(Y))
            Y = times_xc(Z)
        Qy, _ = torch.linalg.qr(Y)
        B = times_xct(Qy).T  # [r, G] = Qy^T Xc
        _, _, Vt = torch.linalg.svd(B, full_matrices=False)
        V = _flip_signs(Vt[:k].T.contiguous())

    Y = torch.empty(N, k, dtype=torch.float64, device=dev)
    for lo, hi in _row_chunks(N, G):
        Y[lo:hi] = (X[lo:hi] - mean) @ V
    if return_model:
        return Y, mean, V
    return Y
