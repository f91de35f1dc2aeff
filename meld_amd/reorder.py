"""Cache-locality permutation of the cells (no reference counterpart; see csrc/reorder.hip).

``locality_permutation(X)`` returns ``perm`` (int64, device) with ``X_new = X[perm]``: cells are grouped by their nearest of
C1 coarse centroids (random cells, fixed seed), inside a coarse cell by their nearest of F1 sub-centroids, and once more by
F2; the centroids of every group are ordered by a greedy nearest-neighbour chain, so that consecutive groups are close in
space.  The fan-outs are chosen for leaves of ~16 cells ((64; 32, 32) at 1M cells): a 64-cell reference tile of the search is
then four neighbouring leaves -- what the pruning bounds of the search pay for -- and a row block of the recurrence kernel
gathers mostly from its own part of the iterate.  One 32-bit key per cell (the position of its group along the chains) is the
group id of the next level; a level is six launches (``meld_assign_nearest`` on the matrix pipe, the chain beside it,
``meld_order_update_keys``, ``meld_argsort_u32``, ``meld_order_starts``, ``meld_order_pick_centroids``).
The graph, and therefore every result, is independent of the order (tests compare against the oracle in the original order);
only memory locality and the cost of the search change.
"""
from __future__ import annotations

from ._options import is_set, opt

import math
import os

import numpy as np
import torch

from ._lib import check, get_lib, ptr

__all__ = ["locality_permutation"]


class _Sorter:
    """Stable argsort of the 32-bit ordering keys (``meld_argsort_u32``: the index payload comes from a counting iterator,
    the sorted keys are kept for the group boundaries); buffers allocated once per permutation."""

    def __init__(self, n, dev):
        self.lib, self.n = get_lib(), n
        self.order = torch.empty(n, dtype=torch.int64, device=dev)
        self.skeys = torch.empty(n, dtype=torch.int32, device=dev)
        self.tb = self.lib.meld_argsort_u32_temp_bytes(n)
        self.tmp = torch.empty(self.tb, dtype=torch.uint8, device=dev)

    def __call__(self, key, n_groups, st):
        bits = int(max(1, int(n_groups - 1).bit_length()))
        check(self.lib.meld_argsort_u32(ptr(key), self.n, bits, ptr(self.order), ptr(self.skeys), ptr(self.tmp), self.tb, st), "meld_argsort_u32")
        return self.order


_SIDE = {}


def _side_stream(dev):
    key = (dev.type, dev.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def locality_permutation(X, c1=None, fanouts=None, seed=0, comm=None):
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

    if opt("MELD_REORDER"):  # tuning hook: "c1,f1,f2,..."
        parts = [int(v) for v in opt("MELD_REORDER").split(",")]
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
    cents = X.index_select(0, idx1).contiguous()
    i32 = dict(dtype=torch.int32, device=dev)
    child = torch.empty(N, **i32)
    key = torch.zeros(N, **i32)  # position of the cell's group along the chains of the levels so far (the group id of the next level)
    main, side = torch.cuda.current_stream(), _side_stream(dev)
    sorter = _Sorter(N, dev)
    # Every level: the chain over the group's centroids (one wave per group, a latency-bound walk of <= 64 greedy steps)
    # does not depend on the assignment of the cells and runs beside it on a second stream; then
    # key <- key * fanout + position of the cell's centroid along its group's chain.  Level 0 is one group of c1 random
    # cells; each further level takes `fanout` evenly spaced members of every group (in sorted order) as sub-centroids.
    # All of it is a handful of launches per level (meld_order_*): issued as tensor operations the stage was host-bound.
    n_groups, f, order = 1, c1, None
    for nxt in tuple(fanouts) + (None,):
        rank = torch.empty(n_groups * f, **i32)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if f > 2:
                check(lib.meld_chain_order(ptr(cents), n_groups, f, d, ptr(rank), side.cuda_stream), "meld_chain_order")
            else:
                rank.copy_(torch.arange(f, **i32).repeat(n_groups))
        if comm is None or comm.world == 1:
            check(lib.meld_assign_nearest(ptr(X), N, d, ptr(cents), f, ptr(key) if order is not None else None,
                                          ptr(order) if order is not None else None, ptr(child), st), "meld_assign_nearest")
        else:
            # row-sharded driver: the assignment -- the one pass over all cells of a level -- is split by position, every
            # rank takes 1 / world of the cells (in the level's traversal order) and the children are all-gathered
            # (4 B per cell); chains, sorts and picks are small and stay replicated, so every rank ends with the same keys
            per = -(-N // comm.world)
            p0 = min(comm.rank * per, N)
            cnt = min(per, N - p0)
            mine = torch.zeros(per, **i32)
            if cnt > 0:
                if order is None:
                    check(lib.meld_assign_nearest(ptr(X[p0:]), cnt, d, ptr(cents), f, None, None, ptr(mine), st), "meld_assign_nearest")
                else:
                    sl = order[p0 : p0 + cnt]
                    check(lib.meld_assign_nearest(ptr(X), cnt, d, ptr(cents), f, ptr(key), ptr(sl), ptr(child), st), "meld_assign_nearest")
                    mine[:cnt] = child.index_select(0, sl)
            everyone = torch.empty(per * comm.world, **i32)
            comm.all_gather_rows(everyone, mine)
            if order is None:
                child.copy_(everyone[:N])
            else:
                child.index_copy_(0, order, everyone[:N])
        main.wait_stream(side)
        check(lib.meld_order_update_keys(ptr(key), ptr(child), ptr(rank), N, f, st), "meld_order_update_keys")
        n_groups *= f
        order = sorter(key, n_groups, st)
        if nxt is None or N // (n_groups * nxt) < 4:
            break
        f = nxt
        starts = torch.empty(n_groups + 1, dtype=torch.int64, device=dev)
        check(lib.meld_order_starts(ptr(sorter.skeys), N, n_groups, ptr(starts), st), "meld_order_starts")
        cents = torch.empty(n_groups * f, d, dtype=torch.float64, device=dev)
        check(lib.meld_order_pick_centroids(ptr(X), N, d, ptr(order), ptr(starts), n_groups, f, ptr(cents), st), "meld_order_pick_centroids")
    return order
