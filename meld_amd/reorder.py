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

import math
import os

import numpy as np
import torch

from ._lib import check, get_lib, ptr

__all__ = ["locality_permutation"]


def _argsort_bits(keys, n_bits):
    """Stable argsort of non-negative int64 keys below 2**n_bits: the library's radix sort over just those bits
    (a 14-bit key takes two passes; torch.argsort sorts all 64)."""
    lib = get_lib()
    n = int(keys.shape[0])
    dev = keys.device
    keys = keys.contiguous()
    idx = torch.arange(n, dtype=torch.float64, device=dev)  # (exact up to 2**53)
    k2, v2 = torch.empty_like(keys), torch.empty_like(idx)
    tb = lib.meld_sort_temp_bytes(n)
    tmp = torch.empty(tb, dtype=torch.uint8, device=dev)
    check(lib.meld_sort_pairs_u64_f64(ptr(keys), ptr(k2), ptr(idx), ptr(v2), n, int(max(1, n_bits)), ptr(tmp), tb,
                                      torch.cuda.current_stream().cuda_stream), "meld_sort_pairs_u64_f64")
    return v2.to(torch.int64)


def _chain_order_batched(P):
    """Greedy nearest-neighbour chain over the rows of every P[b] (CUDA fp64 [B, m, d], m <= 64)
    -> rank [B, m] int64: position of each row along its chain (one wave per group on the device)."""
    B, m, d = P.shape
    dev = P.device
    if m <= 2:
        return torch.arange(m, device=dev, dtype=torch.int64)[None, :].repeat(B, 1)
    P = P.contiguous()
    rank = torch.empty((B, m), dtype=torch.int32, device=dev)
    check(get_lib().meld_chain_order(ptr(P), B, m, d, ptr(rank), torch.cuda.current_stream().cuda_stream), "meld_chain_order")
    return rank.to(torch.int64)


def _chain_order(P):
    return _chain_order_batched(P[None])[0]


_SIDE = {}


def _side_stream(dev):
    key = (dev.type, dev.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def _split_level(X, lib, st, group, n_groups, fanout):
    """One refinement level: every group of cells gets `fanout` sub-centroids (evenly spaced
    members), every cell its nearest one.  Returns (child id within group [N] int64, rank of each
    child along its group's chain [n_groups, fanout])."""
    N, d = int(X.shape[0]), int(X.shape[1])
    dev = X.device
    order = _argsort_bits(group, int(n_groups - 1).bit_length())
    counts = torch.bincount(group, minlength=n_groups)
    starts = torch.cumsum(counts, 0) - counts
    frac = (torch.arange(fanout, device=dev, dtype=torch.float64) + 0.5) / fanout
    pick = starts[:, None] + (frac[None, :] * counts[:, None].to(torch.float64)).to(torch.int64)
    pick = torch.minimum(pick, (starts + torch.clamp(counts - 1, min=0))[:, None]).clamp_(0, N - 1)
    cents = X.index_select(0, order[pick.reshape(-1)]).contiguous()  # [n_groups * fanout, d]
    child = torch.empty(N, dtype=torch.int32, device=dev)
    g32 = group.to(torch.int32)
    # the chains over the sub-centroids (one wave per group, a latency-bound walk of <= 64 greedy steps: 0.14-0.27 ms)
    # do not depend on the assignment of the cells: they run beside it on a second stream
    main = torch.cuda.current_stream()
    side = _side_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        rank = _chain_order_batched(cents.reshape(n_groups, fanout, d))
    check(lib.meld_assign_nearest(ptr(X), N, d, ptr(cents), fanout, ptr(g32), ptr(order), ptr(child), st), "meld_assign_nearest")
    main.wait_stream(side)
    return child.to(torch.int64), rank


def locality_permutation(X, c1=None, fanouts=None, seed=0):
    """X: CUDA fp64 [N, d].  Returns perm (device int64 [N]) or None when N is too small to matter.

    Level 0: nearest of c1 (<= 64) random cells (coarse cells, ordered by a chain); each further
    level splits every group into `fanout` children by nearest sub-centroid, while leaves keep >= 4
    cells on average.  Measured at 1M cells (Chebyshev step / cost of building the permutation):
    (64;16) 226 us / 12 ms, (64;16,16) 209 us / 19 ms, (16;16,16,16) 204 us / 24 ms, none 580 us --
    finer leaves buy little in a 10-d intrinsic geometry, so the default stops at ~1000-cell leaves: consecutive rows are mutual near neighbours, and a
    128-byte line of the iterate (8 rows) holds cells that the same row block gathers again."""
    lib = get_lib()
    N, d = int(X.shape[0]), int(X.shape[1])
    if N < 8192 or d > 128:  # (the assignment kernels stage <= 128 coordinates; wide data keeps its order)
        return None
    import os

    if os.environ.get("MELD_REORDER"):  # tuning hook: "c1,f1,f2,..."
        parts = [int(v) for v in os.environ["MELD_REORDER"].split(",")]
        c1, fanouts = parts[0], tuple(parts[1:])
    st = torch.cuda.current_stream().cuda_stream
    dev = X.device
    if c1 is None:
        c1 = int(min(64, max(8, N // 4096)))
    if fanouts is None:
        # leaves of ~16 cells: a 64-cell reference tile of the search is then four neighbouring leaves and its radius --
        # what the per-query pruning test (meld_knn16_bounds) pays for -- stays small.  Measured with the seeded pruning
        # table, whole step: 500k cells (64;16,32) 30.4 ms vs 33.9 with (64;16,16); 1M (64;32,32) 66.9 vs 69.5; 2M
        # (64;64,32) 185 vs 226; finer still costs more in the ordering than it saves.
        shapes = ((16, 16), (16, 32), (32, 32), (64, 32), (64, 64))
        fanouts = min(shapes, key=lambda f: abs(math.log(c1 * f[0] * f[1] / max(N / 16.0, 1.0))))
    rng = np.random.default_rng(seed)
    idx1 = torch.from_numpy(np.sort(rng.choice(N, size=c1, replace=False))).to(dev)
    cents1 = X.index_select(0, idx1).contiguous()
    a1 = torch.empty(N, dtype=torch.int32, device=dev)
    main, side = torch.cuda.current_stream(), _side_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        rank1 = _chain_order(cents1)  # (one wave, 63 dependent steps: hidden behind the assignment)
    check(lib.meld_assign_nearest(ptr(X), N, d, ptr(cents1), c1, None, None, ptr(a1), st), "meld_assign_nearest")
    main.wait_stream(side)
    key = rank1[a1.to(torch.int64)]  # order of the coarse cell of every point
    group = a1.to(torch.int64)
    n_groups = c1
    for f in fanouts:
        if N // (n_groups * f) < 4:
            break
        child, rank = _split_level(X, lib, st, group, n_groups, f)
        key = key * f + rank[group, child]
        group = group * f + child
        n_groups *= f
    return _argsort_bits(key, int(n_groups - 1).bit_length())
