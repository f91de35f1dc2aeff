"""Cache-locality permutation of the cells (no reference counterpart; see csrc/reorder.hip).

``locality_permutation(X)`` returns ``perm`` (int64, device) with ``X_new = X[perm]``: cells are
grouped by their nearest of C1 coarse centroids (random cells, fixed seed) and, inside a coarse
cell, by their nearest of C2 sub-centroids; centroids are ordered by a greedy nearest-neighbour
chain so that consecutive groups are close in space.  Leaves hold ~N / (C1*C2) cells (~250 at 1M),
so a row block of the recurrence kernel gathers mostly from its own few KiB of the iterate.
The graph, and therefore every result, is independent of the order (tests compare against the
oracle in the original order); only memory locality changes.
"""
from __future__ import annotations

import numpy as np
import torch

from ._lib import check, get_lib, ptr

__all__ = ["locality_permutation"]


def _chain_order_batched(P):
    """Greedy nearest-neighbour chain over the rows of every P[b] ([B, m, d]) -> rank [B, m]
    (rank of each row along its chain).  Vectorised over the batch: m numpy steps."""
    B, m, _ = P.shape
    rank = np.zeros((B, m), dtype=np.int64)
    if m <= 2:
        rank[:] = np.arange(m)
        return rank
    D = ((P[:, :, None, :] - P[:, None, :, :]) ** 2).sum(-1)  # [B, m, m]
    ar = np.arange(B)
    cur = np.argmin(P[:, :, 0], axis=1)  # start from an extreme point along the first coordinate
    used = np.zeros((B, m), dtype=bool)
    used[ar, cur] = True
    rank[ar, cur] = 0
    for step in range(1, m):
        row = np.where(used, np.inf, D[ar, cur])
        cur = np.argmin(row, axis=1)
        used[ar, cur] = True
        rank[ar, cur] = step
    return rank


def _chain_order(P):
    return _chain_order_batched(P[None])[0]


def locality_permutation(X, c1=None, c2=16, seed=0):
    """X: CUDA fp64 [N, d].  Returns perm (device int64 [N]) or None when N is too small to matter."""
    lib = get_lib()
    N, d = int(X.shape[0]), int(X.shape[1])
    if N < 8192:
        return None
    st = torch.cuda.current_stream().cuda_stream
    dev = X.device
    if c1 is None:
        c1 = int(min(1024, max(16, N // 4096)))  # ~256 leaves' worth of cells per coarse cell at most
    rng = np.random.default_rng(seed)
    idx1 = torch.from_numpy(np.sort(rng.choice(N, size=c1, replace=False))).to(dev)
    cents1 = X.index_select(0, idx1).contiguous()
    a1 = torch.empty(N, dtype=torch.int32, device=dev)
    check(lib.meld_assign_nearest(ptr(X), N, d, ptr(cents1), c1, None, ptr(a1), st), "meld_assign_nearest")
    rank1 = torch.from_numpy(_chain_order(cents1.cpu().numpy())).to(dev)

    # sub-centroids: c2 evenly spaced members of every coarse cell (cells sorted by coarse id)
    order1 = torch.argsort(a1.to(torch.int64), stable=True)
    counts = torch.bincount(a1.to(torch.int64), minlength=c1)
    starts = torch.cumsum(counts, 0) - counts
    frac = (torch.arange(c2, device=dev, dtype=torch.float64) + 0.5) / c2
    pick = starts[:, None] + torch.clamp((frac[None, :] * counts[:, None].to(torch.float64)).to(torch.int64), max=N - 1)
    pick = torch.minimum(pick, (starts + torch.clamp(counts - 1, min=0))[:, None])  # empty cells cannot occur (own centroid)
    sub_idx = order1[pick.reshape(-1)]
    cents2 = X.index_select(0, sub_idx).contiguous()  # [c1 * c2, d]
    a2 = torch.empty(N, dtype=torch.int32, device=dev)
    check(lib.meld_assign_nearest(ptr(X), N, d, ptr(cents2), c2, ptr(a1), ptr(a2), st), "meld_assign_nearest")
    c2h = cents2.cpu().numpy().reshape(c1, c2, d)
    rank2 = torch.from_numpy(_chain_order_batched(c2h)).to(dev)  # [c1, c2]

    a1l, a2l = a1.to(torch.int64), a2.to(torch.int64)
    key = rank1[a1l] * c2 + rank2[a1l, a2l]
    return torch.argsort(key, stable=True)
