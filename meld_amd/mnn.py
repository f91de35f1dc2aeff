"""Mutual-nearest-neighbours graph between samples (``sample_idx``).

``meld.MELD().fit_transform(data, labels, sample_idx=labels)`` forwards ``sample_idx`` to
``graphtools.Graph`` (reference ``meld/meld.py:117-118``; exercised by ``test/test_meld.py:34`` and
``test/test_utils.py:11``), which then builds its ``MNNGraph`` [UPSTREAM graphtools 1.5.x
``MNNGraph.build_kernel``]:

* the block of every sample with itself is that sample's own symmetrised alpha-decay kernel
  ``(k + k^T) / 2`` -- built here by the hot path itself (``HipOps.directed_kernel_coo`` on the sample's
  cells: MFMA candidate search, exact refinement);
* the block from sample i to another sample j is the alpha-decay kernel from i's cells to j's cells with the
  bandwidth at the knn-th nearest cell *of j* (no self among the references), each row scaled by
  ``min(1, within / between) * beta`` (row sum of the diagonal block over row sum of this block) -- built
  by the same hot path with the references restricted to sample j (``_cross_block``: the MFMA search takes the
  queries and the references from different row ranges; no pruning table or seeds there, they rest on tile = query
  block).  ``cross_kernel`` (chunked fp64 rocBLAS GEMMs + ``topk``) remains for data wider than the search kernels;
* the assembled matrix goes through the common symmetrise / anisotropy / degree steps (``csrc/assemble.hip``).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .graph import DeviceGraph, HipOps, resolve_graph_params

__all__ = ["build_mnn_graph", "cross_kernel"]


def _exact_dist(Xq, Yr, qi, ri):
    """Euclidean distances of the pairs (qi[k], ri[k]) by direct differences (fp64)."""
    out = torch.empty(qi.shape[0], dtype=torch.float64, device=Xq.device)
    step = 1 << 22
    for s in range(0, qi.shape[0], step):
        diff = Xq.index_select(0, qi[s:s + step]) - Yr.index_select(0, ri[s:s + step])
        out[s:s + step] = torch.sqrt((diff * diff).sum(dim=1))
    return out


def cross_kernel(Xq, Yr, knn, decay, thresh, q_chunk=4096, r_chunk=32768):
    """Directed alpha-decay kernel from the rows of ``Xq`` to the rows of ``Yr`` (different point sets):
    (row, col, value) of every entry with value >= thresh, bandwidth = distance to the knn-th nearest row of
    ``Yr``.  ``decay = inf``: the unweighted kernel (1 for the knn nearest)."""
    dev = Xq.device
    nq, nr = int(Xq.shape[0]), int(Yr.shape[0])
    knn = int(min(knn, nr))  # graphtools clips knn to the size of the reference sample
    mean = Yr.mean(dim=0)
    Xq = (Xq - mean).contiguous()
    Yr = (Yr - mean).contiguous()
    n2r = (Yr * Yr).sum(dim=1)
    rf = 1.0 if math.isinf(decay) else float((-math.log(thresh)) ** (1.0 / decay))
    kk = min(nr, knn + 8)  # a few spares: the GEMM-form distances only rank the candidates
    rows, cols, vals = [], [], []
    for q0 in range(0, nq, q_chunk):
        q1 = min(nq, q0 + q_chunk)
        Q = Xq[q0:q1]
        n2q = (Q * Q).sum(dim=1)
        m = q1 - q0
        best_d = torch.empty((m, 0), dtype=torch.float64, device=dev)
        best_i = torch.empty((m, 0), dtype=torch.int64, device=dev)
        for r0 in range(0, nr, r_chunk):
            r1 = min(nr, r0 + r_chunk)
            D = n2q[:, None] + n2r[None, r0:r1] - 2.0 * (Q @ Yr[r0:r1].T)
            ids = torch.arange(r0, r1, device=dev, dtype=torch.int64)[None, :].expand(m, -1)
            D = torch.cat([best_d, D], dim=1)
            ids = torch.cat([best_i, ids], dim=1)
            best_d, sel = torch.topk(D, min(kk, D.shape[1]), dim=1, largest=False, sorted=False)
            best_i = torch.gather(ids, 1, sel)
        kcur = best_i.shape[1]
        qrep = torch.arange(q0, q1, device=dev, dtype=torch.int64)[:, None].expand(-1, kcur).reshape(-1)
        dex = _exact_dist(Xq, Yr, qrep, best_i.reshape(-1)).reshape(m, kcur)
        dsort, order = torch.sort(dex, dim=1)
        bw = torch.clamp(dsort[:, knn - 1], min=float(np.finfo(float).eps))
        if math.isinf(decay):
            near = torch.gather(best_i, 1, order[:, :knn])
            rows.append(torch.arange(q0, q1, device=dev, dtype=torch.int64)[:, None].expand(-1, knn).reshape(-1))
            cols.append(near.reshape(-1))
            vals.append(torch.ones(m * knn, dtype=torch.float64, device=dev))
            continue
        rad2 = (bw * rf) ** 2
        slack = 1e-9 * rad2 + 1e-12 * (n2q + n2r.max())  # GEMM-form rounding: the exact test follows
        for r0 in range(0, nr, r_chunk):
            r1 = min(nr, r0 + r_chunk)
            D = n2q[:, None] + n2r[None, r0:r1] - 2.0 * (Q @ Yr[r0:r1].T)
            hit = torch.nonzero(D <= (rad2 + slack)[:, None])
            if hit.shape[0] == 0:
                continue
            qi = hit[:, 0] + q0
            ri = hit[:, 1] + r0
            dist = _exact_dist(Xq, Yr, qi, ri)
            v = torch.exp(-torch.pow(dist / bw[hit[:, 0]], decay))
            v = torch.where(torch.isnan(v), torch.ones_like(v), v)
            keep = v >= thresh
            rows.append(qi[keep])
            cols.append(ri[keep])
            vals.append(v[keep])
    if not rows:
        z = torch.empty(0, dtype=torch.int64, device=dev)
        return z, z.clone(), torch.empty(0, dtype=torch.float64, device=dev)
    return torch.cat(rows), torch.cat(cols), torch.cat(vals)


def _cross_block(ops, Xq, Yr, knn, decay, thresh, ksel=None):
    """Directed kernel from the cells of one sample (``Xq``) to the cells of another (``Yr``) on the hot path: the MFMA
    candidate search of ``csrc/knn16.hip`` with the references restricted to ``Yr`` (queries stacked behind them,
    ``HipOps.directed_kernel_coo(n_refs=)``), exact refinement, certification / exact sweep as for a sample's own block.
    There is no self among a query's candidates, so the bandwidth is its knn-th -- not (knn+1)-th -- nearest reference.
    Data wider than the search kernels (d > 141) keeps the library path (``cross_kernel``)."""
    from .graph import default_ksel

    nq, nr, d = int(Xq.shape[0]), int(Yr.shape[0]), int(Xq.shape[1])
    knn_c = int(min(knn, nr))  # graphtools clips knn to the size of the reference sample
    if ops.search != "f16x3" or ops.lib.meld_knn16_kblocks(d) < 0 or knn_c < 2 or nr < 3:
        return cross_kernel(Xq, Yr, knn, decay, thresh)
    Xcat = torch.cat([Yr, Xq], dim=0).contiguous()
    ks = int(ksel) if ksel is not None else default_ksel(knn_c)
    keys, vals, _, _ = ops.directed_kernel_coo(Xcat, nr, nq, knn_c - 1, decay, thresh, ks, n_refs=nr)
    M = keys.shape[0] // 2
    return (keys[:M] >> 32) - nr, keys[:M] & 0xFFFFFFFF, 2.0 * vals[:M]  # (the COO stream carries K / 2)


def build_mnn_graph(X, sample_idx, knn=5, decay=40, thresh=1e-4, anisotropy=1, beta=1.0, ksel=None):
    """Data [N, d] (CUDA fp64) + per-cell sample labels -> DeviceGraph of the MNN kernel (cells keep their order)."""
    if not (isinstance(X, torch.Tensor) and X.is_cuda and X.dtype == torch.float64 and X.dim() == 2):
        raise TypeError("build_mnn_graph expects a CUDA float64 tensor [N, d]")
    N, d = int(X.shape[0]), int(X.shape[1])
    sample_idx = np.asarray(sample_idx)
    if sample_idx.ndim != 1 or sample_idx.shape[0] != N:
        raise ValueError("sample_idx ({}) must be the same length as data ({})".format(sample_idx.shape[0], N))
    samples, codes = np.unique(sample_idx, return_inverse=True)
    if len(samples) == 1:
        raise ValueError("sample_idx must contain more than one unique value")
    dev = X.device
    ops = HipOps(dev)
    thresh = float(max(thresh, np.finfo(float).eps))
    members = [torch.from_numpy(np.nonzero(codes == s)[0]).to(dev) for s in range(len(samples))]
    for s, m in zip(samples, members):
        if m.shape[0] < 3:
            raise ValueError("sample {!r} has {} cells; every sample needs at least 3".format(s, int(m.shape[0])))
    parts = [X.index_select(0, m).contiguous() for m in members]
    k_rows, k_cols, k_vals = [], [], []
    within = []
    n_flagged = 0
    for mi, Xi in zip(members, parts):
        n_i = int(Xi.shape[0])
        knn_i, thresh_i, ksel_i = resolve_graph_params(n_i, knn, thresh, ksel)
        keys, vals, _, info = ops.directed_kernel_coo(Xi, 0, n_i, knn_i, decay, thresh_i, ksel_i)
        n_flagged += int(info.get("n_flagged_rows", 0))
        M = keys.shape[0] // 2
        r, c, hv = keys[:M] >> 32, keys[:M] & 0xFFFFFFFF, vals[:M]  # hv = k_rc / 2
        w = torch.ones(n_i, dtype=torch.float64, device=dev)  # row sums of (k + k^T)/2, diagonal 1 included
        w.index_add_(0, r, hv)
        w.index_add_(0, c, hv)
        within.append(w)
        gr, gc = mi[r], mi[c]
        k_rows += [gr, gc]
        k_cols += [gc, gr]
        k_vals += [hv, hv]
    for i, (mi, Xi) in enumerate(zip(members, parts)):
        for j, (mj, Xj) in enumerate(zip(members, parts)):
            if i == j:
                continue
            r, c, v = _cross_block(ops, Xi, Xj, knn, decay, thresh, ksel)
            between = torch.zeros(int(Xi.shape[0]), dtype=torch.float64, device=dev)
            between.index_add_(0, r, v)
            scale = torch.clamp(within[i] / between, max=1.0) * float(beta)
            hv = 0.5 * v * scale[r]
            gr, gc = mi[r], mj[c]
            k_rows += [gr, gc]
            k_cols += [gc, gr]
            k_vals += [hv, hv]
    keys = (torch.cat(k_rows) << 32) | torch.cat(k_cols)
    vals = torch.cat(k_vals).contiguous()
    if keys.shape[0] == 0:
        raise ValueError("the kernel has no off-diagonal entries; cannot build a graph")
    rowptr, col, val = ops.assemble_rows(keys.contiguous(), vals, 0, N, N)
    ksum = ops.row_sums(rowptr, val, N, 1.0)
    dw = ops.anisotropy_degrees(rowptr, col, val, N, ksum, 0, anisotropy)
    nnz = int(col.shape[0])
    info = dict(N=N, d=d, knn=int(knn), nnz=nnz, mean_degree=nnz / N, graph="mnn", n_samples=len(samples),
                n_flagged_rows=n_flagged, search="f16x3 within and between samples")
    G = DeviceGraph(rowptr, col, val, dw, ksum=ksum, anisotropy=anisotropy, info=info)
    G.ops = ops
    return G
