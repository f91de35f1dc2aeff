"""Row-sharded MELD across the GPUs of one node (one process per GPU, ``torch.distributed``;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no distributed code; this is the design of SURVEY.md section 8(e):

* cells are row-sharded: rank g owns rows [g*R, (g+1)*R), R = ceil(N / world) (the tail is padded
  with isolated rows so that every collective is equal-sized);
* kNN: X is replicated; each rank searches its own queries against all N references -- no
  communication in the dominant, compute-bound stage;
* symmetrisation (K + K^T)/2: every directed entry (i -> j) also belongs to row j's owner, so the
  transposed COO entries are bucketed by owner and exchanged with ONE all-to-all-v; each rank then
  sort-merges its own rows;
* anisotropy needs the kernel row sums of remote columns: one all-gather of an [N] fp64 vector;
* lmax: Lanczos with the sharded SpMV (all-gather of the iterate + two scalar all-reduces per
  iteration);
* Chebyshev: each step computes the local rows of T_k and all-gathers the [N, p] iterate.

Local work goes through an ``ops`` object (``meld_amd.graph.HipOps`` in production); the CPU
tests inject a NumPy stand-in to exercise exactly this communication code under gloo.
"""
from __future__ import annotations

from ._options import is_set, opt

import os

import numpy as np
import pandas as pd
import torch
import torch.distributed as dist

from .graph import DeviceGraph, resolve_graph_params

__all__ = ["Comm", "shard_range", "build_sharded_graph", "shard_of_graph", "fit_transform_sharded"]


class Comm:
    """Thin wrapper over the default process group (equal-sized collectives only, plus one
    all-to-all-v)."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def rccl(self):
        """Handle of the library's own RCCL communicator over this group (``meld_rccl_comm_create``), for the C-side recurrence
        loops (``meld_cheby_run_sharded`` / ``meld_lanczos_steps_sharded``: kernel and collective of every step enqueued from
        one call), or None when the group does not run on RCCL (gloo, the host-staged test collectives) or the loops are
        switched off (``MELD_SHARDED_C_LOOPS=0``).  The unique id travels through torch.distributed.  Created once per process
        group (ncclCommInitRank is a collective); the cache entry holds the group OBJECT, so a group torn down and re-created --
        or another one at a recycled ``id()`` -- never sees a stale handle."""
        try:
            pg = self.group if self.group is not None else dist.distributed_c10d._get_default_group()
        except Exception:
            pg = self.group
        hit = Comm._RCCL.get(id(pg))
        if hit is not None and hit[0] is pg:
            return hit[1]
        handle = None
        try:
            eligible = dist.get_backend(self.group) == "nccl" and torch.cuda.is_available() and opt("MELD_SHARDED_C_LOOPS", "1") != "0"
        except Exception:
            eligible = False
        if eligible:
            import ctypes as C

            from ._lib import check, get_lib

            lib = get_lib()
            # (every rank takes the same branch: the library and its RCCL are the same build on all of them)
            if lib.meld_rccl_available():
                buf = C.create_string_buffer(128)
                if self.rank == 0:
                    check(lib.meld_rccl_unique_id(buf), "meld_rccl_unique_id")
                box = [bytes(buf.raw)]
                src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
                dist.broadcast_object_list(box, src=src, group=self.group)
                h = C.c_void_p()

                def agreed(ok):  # every rank takes the same path: MIN over the group, through torch.distributed
                    flag = torch.tensor([int(ok)], dtype=torch.int32, device="cuda")
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
                    return int(flag.item()) == 1

                try:
                    check(lib.meld_rccl_comm_create(box[0], self.world, self.rank, C.byref(h)), "meld_rccl_comm_create")
                    created = True
                except Exception:
                    created = False
                # agreement on the creation BEFORE any collective is issued on the new communicator: a rank whose create failed
                # would leave the others waiting inside the probe below
                ok = agreed(created)
                if ok:
                    try:
                        # one all-reduce and one in-place all-gather through the new communicator before it is trusted with the filter
                        st = torch.cuda.current_stream().cuda_stream
                        probe = torch.ones(4, dtype=torch.float64, device="cuda")
                        check(lib.meld_rccl_all_reduce_sum_f64(h, C.c_void_p(probe.data_ptr()), 4, st), "meld_rccl_all_reduce_sum_f64")
                        full = torch.full((self.world, 2), -1.0, dtype=torch.float64, device="cuda")
                        full[self.rank] = float(self.rank)
                        mine = full[self.rank]
                        check(lib.meld_rccl_all_gather(h, C.c_void_p(mine.data_ptr()), C.c_void_p(full.data_ptr()), 16, st), "meld_rccl_all_gather")
                        want = torch.arange(self.world, dtype=torch.float64, device="cuda")[:, None].expand(-1, 2)
                        good = bool((probe == float(self.world)).all()) and bool((full == want).all())
                    except Exception:
                        good = False
                    ok = agreed(good)
                if ok:
                    handle = h
                else:
                    if h.value:
                        lib.meld_rccl_comm_destroy(h)
                    import warnings

                    warnings.warn("meld_amd: the library's own RCCL communicator failed its self-check; the sharded recurrences run their "
                                  "per-step loops over torch.distributed", RuntimeWarning)
        Comm._RCCL[id(pg)] = (pg, handle)  # (the strong reference keeps the id from being recycled while the entry lives)
        return handle

    _RCCL = {}  # id(process group) -> (the group object, the library's RCCL communicator on it or None)

    @classmethod
    def release_rccl(cls):
        """Destroy the cached communicators (call before ``dist.destroy_process_group()`` in a long-lived process)."""
        from ._lib import get_lib

        for _, h in cls._RCCL.values():
            if h is not None and h.value:
                get_lib().meld_rccl_comm_destroy(h)
        cls._RCCL.clear()

    def all_gather_rows(self, full, local):
        """full[rank*R:(rank+1)*R] <- local, for every rank (in place when local is that slice)."""
        dist.all_gather_into_tensor(full, local.contiguous(), group=self.group)

    def all_reduce_sum(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_max(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def exchange_fixed(self, send, cap):
        """ONE equal-split all-to-all of the fixed-capacity send buffer [world, 2, cap] (``ops.partition_remote``): no
        split sizes, hence no host read-back.  Returns (keys, vals) of world * cap slots each; unused slots carry the
        sentinel key ~0, which the assembly ignores."""
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)
        r = recv.view(self.world, 2, cap)
        return r[:, 0, :].reshape(-1), r[:, 1, :].reshape(-1).view(torch.float64)

    def exchange_by_owner(self, keys_sorted, vals_sorted, rows_per_rank):
        """keys are (row << 32 | col), sorted; entry e goes to rank row // rows_per_rank.
        Returns the (keys, vals) received from every rank (concatenated)."""
        dev = keys_sorted.device
        bounds = (torch.arange(1, self.world, dtype=torch.int64, device=dev) * rows_per_rank) << 32
        cuts = torch.searchsorted(keys_sorted, bounds)
        edges = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), cuts, torch.tensor([keys_sorted.shape[0]], dtype=torch.int64, device=dev)])
        counts = torch.empty(2, self.world, dtype=torch.int64, device=dev)  # [0] what I send, [1] what I receive
        counts[0] = edges[1:] - edges[:-1]
        dist.all_to_all_single(counts[1], counts[0], group=self.group)
        # the split sizes of the variable-length exchange have to be host integers: ONE read-back for both
        counts_h = counts.cpu().tolist()
        send_l, recv_l = counts_h[0], counts_h[1]
        n_recv = sum(recv_l)
        rk = torch.empty(n_recv, dtype=keys_sorted.dtype, device=dev)
        rv = torch.empty(n_recv, dtype=vals_sorted.dtype, device=dev)
        dist.all_to_all_single(rk, keys_sorted.contiguous(), output_split_sizes=recv_l, input_split_sizes=send_l, group=self.group)
        dist.all_to_all_single(rv, vals_sorted.contiguous(), output_split_sizes=recv_l, input_split_sizes=send_l, group=self.group)
        return rk, rv


def exchange_capacity(rows_per_rank, ksel, world, locality=False):
    """Slots per peer of the fixed-capacity exchange of transposed entries.  A rank emits at most rows_per_rank * ksel
    directed entries, in practice about a third of that (the candidates that survive the kernel threshold: ~20 per row at
    knn = 15), each owed to the owner of its column.  With the cells in locality order most of them stay on their own rank
    and never enter the exchange (measured with tools/shard_emulate.py: a 1/8 shard of the 1M benchmark emits 3.3 M
    entries and owes 11 % of them to other ranks, at most 141 k = 4 % to any one): a quarter of the even share of the bound
    (250 k at 8 ranks) covers that and keeps the buffer -- memset, all-to-all and the scatter's scan of it -- small (64 MB
    instead of 256 MB per rank at 8 ranks); without an ordering the entries spread evenly and the capacity is the even
    share of the bound.  An overflow is detected and falls back to the variable-length exchange.
    ``MELD_EXCHANGE_CAP`` overrides (the tests force the fallback with it)."""
    env = opt("MELD_EXCHANGE_CAP")
    if env:
        return int(env)
    total = int(rows_per_rank) * int(ksel)
    share = total // max(world, 1)
    cap = min(total, (share // 4 if locality else share) + 1024)
    return ((cap + 255) // 256) * 256


def shard_range(N, world, rank):
    """(rows_per_rank R, first row, number of real rows) of a rank."""
    R = (N + world - 1) // world
    R = ((R + 255) // 256) * 256  # whole search workgroups: tile-aligned shards keep the pruning tables usable
    begin = min(rank * R, N)
    end = min(begin + R, N)
    return R, begin, end - begin


def build_sharded_graph(X, ops, comm, knn=5, decay=40, thresh=1e-4, anisotropy=1, ksel=None, reorder=True):
    """Every rank holds the full ``X`` [N, d] (fp64, on its device) and builds the rows it owns.
    With ``reorder`` every rank computes the same (deterministic) locality permutation and shards
    the permuted cells, so a rank's rows are spatially coherent."""
    N, d = int(X.shape[0]), int(X.shape[1])
    knn, thresh, ksel = resolve_graph_params(N, knn, thresh, ksel)
    perm = None
    if reorder and X.is_cuda:
        from .reorder import locality_permutation

        perm = locality_permutation(X, comm=comm)  # (the assignment passes are split over the ranks)
        if perm is not None:
            X = ops.gather_rows(X, perm) if hasattr(ops, "gather_rows") else X.index_select(0, perm)
    R, r0, n_loc = shard_range(N, comm.world, comm.rank)
    dev = X.device

    if n_loc > 0:
        # (collectives inside the build -- the shared tile spheres -- only when EVERY rank builds rows: a rank without rows
        # never gets here; the test is the same on all ranks)
        every_rank_has_rows = (comm.world - 1) * R < N
        extra = {"comm": comm} if (getattr(ops, "shards_spheres", False) and every_rank_has_rows) else {}
        keys, vals, bw, info = ops.directed_kernel_coo(X, r0, n_loc, knn, decay, thresh, ksel, **extra)
    else:
        keys = torch.empty(0, dtype=torch.int64, device=dev)
        vals = torch.empty(0, dtype=torch.float64, device=dev)
        bw, info = torch.empty(0, dtype=torch.float64, device=dev), dict(ksel=ksel, n_flagged_rows=0, nnz_directed=0)
    M = keys.shape[0] // 2
    direct_k, direct_v = keys[:M], vals[:M]  # rows owned by this rank
    trans_k, trans_v = keys[M:].contiguous(), vals[M:].contiguous()  # rows owned by anyone
    rows_here = max(n_loc, 1)
    cap = exchange_capacity(R, ksel, comm.world, locality=perm is not None)

    def assemble(fixed):
        if fixed:
            # the entries owed to other ranks go out in one equal-split all-to-all (capacity `cap` per peer, no split
            # sizes to read back); this rank's own transposed entries need no partition: the assembly ignores rows
            # outside its slice, so the whole array is handed over
            send, counts = ops.partition_remote(trans_k, trans_v, R, comm.world, comm.rank, cap)
            over = (counts.to(torch.int64) - cap).clamp_(min=0).sum().reshape(1)  # entries that did not fit (device)
            recv_k, recv_v = comm.exchange_fixed(send, cap)
            all_k = torch.cat([direct_k, trans_k, recv_k])
            all_v = torch.cat([direct_v, trans_v, recv_v])
            rowptr, col, val = ops.assemble_rows(all_k, all_v, r0, rows_here, N, foreign=True)
        else:
            sk, sv = ops.sort_pairs(trans_k, trans_v, N)
            recv_k, recv_v = comm.exchange_by_owner(sk, sv, R)
            over = torch.zeros(1, dtype=torch.int64, device=dev)
            rowptr, col, val = ops.assemble_rows(torch.cat([direct_k, recv_k]), torch.cat([direct_v, recv_v]), r0, rows_here, N)
        if n_loc == 0:
            rowptr = torch.zeros(2, dtype=torch.int64, device=dev)

        # kernel row sums (diag = K_ii = 1 included) of every row, for the anisotropy of remote columns
        ksum_loc = torch.ones(R, dtype=torch.float64, device=dev)
        if n_loc > 0:
            ksum_loc[:n_loc] = ops.row_sums(rowptr, val, n_loc, 1.0)
        ksum_all = torch.empty(R * comm.world, dtype=torch.float64, device=dev)
        comm.all_gather_rows(ksum_all, ksum_loc)
        if n_loc > 0:
            dw = ops.anisotropy_degrees(rowptr, col, val, n_loc, ksum_all, r0, anisotropy)
        else:
            dw = torch.zeros(1, dtype=torch.float64, device=dev)
        tot = torch.cat([torch.tensor([int(col.shape[0]), int(info["n_flagged_rows"])], dtype=torch.int64, device=dev), over])
        comm.all_reduce_sum(tot)
        return rowptr, col, val, dw, ksum_all, [int(v) for v in tot.tolist()]

    fixed = opt("MELD_EXCHANGE", "fixed") != "variable" and hasattr(ops, "partition_remote") and cap > 0
    rowptr, col, val, dw, ksum_all, (nnz_global, n_flagged, n_over) = assemble(fixed)
    if n_over > 0:  # some rank owed a peer more than the capacity (every rank sees the same total): variable-length exchange
        rowptr, col, val, dw, ksum_all, (nnz_global, n_flagged, _) = assemble(False)
    info.update(N=N, d=d, knn=knn, nnz=int(col.shape[0]), nnz_global=nnz_global, n_flagged_rows=n_flagged,
                rows_per_rank=R, row_begin=r0, rows_local=n_loc, world=comm.world,
                exchange="fixed" if fixed and n_over == 0 else "variable", exchange_capacity=cap, exchange_overflow=n_over)
    G = DeviceGraph(rowptr, col, val, dw, ksum=ksum_all, anisotropy=anisotropy, row_begin=r0, n_total=N, info=info)
    G.n_rows = n_loc
    G.rows_pad = R
    G.n_pad = R * comm.world
    G.comm = comm
    G.ops = ops
    G.bandwidth = bw
    G.perm = perm
    return G


def shard_of_graph(Gf, ops, comm):
    """This rank's row shard of a graph EVERY rank holds in full (built redundantly or adopted): the recurrences -- lmax,
    Chebyshev, the filter-bank VertexFrequencyCluster -- then run row-sharded exactly as on a graph from
    ``build_sharded_graph``.  For the graph kinds the sharded builder does not build itself (the MNN graph of ``sample_idx``,
    a graph built elsewhere): the build is replicated, the filter is sharded."""
    N = int(Gf.N)
    R, r0, n_loc = shard_range(N, comm.world, comm.rank)
    dev = Gf.val.device
    if n_loc > 0:
        e0, e1 = int(Gf.rowptr[r0]), int(Gf.rowptr[r0 + n_loc])
        rowptr = (Gf.rowptr[r0 : r0 + n_loc + 1] - e0).contiguous()
        col, val = Gf.col[e0:e1].contiguous(), Gf.val[e0:e1].contiguous()
        dw = Gf.dw_dev[r0 : r0 + n_loc].contiguous()
    else:  # (a rank beyond the last row: it still takes part in every collective)
        rowptr = torch.zeros(2, dtype=torch.int64, device=dev)
        col = torch.empty(0, dtype=torch.int32, device=dev)
        val = torch.empty(0, dtype=torch.float64, device=dev)
        dw = torch.zeros(1, dtype=torch.float64, device=dev)
    ksum_all = torch.ones(R * comm.world, dtype=torch.float64, device=dev)
    if Gf.ksum is not None:
        ksum_all[:N] = Gf.ksum[:N]
    info = dict(Gf.info)
    info.update(nnz=int(col.shape[0]), nnz_global=int(Gf.nnz), rows_per_rank=R, row_begin=r0, rows_local=n_loc, world=comm.world,
                build="replicated on every rank, rows sharded for the recurrences")
    G = DeviceGraph(rowptr, col, val, dw, ksum=ksum_all, anisotropy=Gf.anisotropy, row_begin=r0, n_total=N, info=info)
    G.n_rows = n_loc
    G.rows_pad = R
    G.n_pad = R * comm.world
    G.comm = comm
    G.ops = ops
    G.bandwidth = getattr(Gf, "bandwidth", None)
    G.perm = Gf.perm
    if getattr(Gf, "_lmax", None) is not None:
        G.lmax = Gf._lmax
    return G


def fit_transform_sharded(op, X, sample_labels, ops=None, comm=None):
    """``op.fit_transform(X, sample_labels)`` with the cells row-sharded over the process group.
    ``op`` is a ``meld_amd.MELD``; every rank passes the same ``X`` / labels and receives the same
    full ``[N, p]`` DataFrame."""
    if comm is None:
        comm = Comm()
    if ops is None:
        from .graph import HipOps

        ops = HipOps()
    if not isinstance(X, torch.Tensor):
        X = torch.from_numpy(np.ascontiguousarray(np.asarray(getattr(X, "values", X)), dtype=np.float64))
    X = X.to(device=ops.device, dtype=torch.float64).contiguous()
    # Graphs the row-sharded BUILDER does not build itself -- graphtools' bandwidth / bandwidth_scale / knn_max / kernel_symm / theta,
    # the dense "exact" graph of thresh = 0, precomputed matrices, the manhattan / chebyshev metrics, knn > 126 -- are built whole on
    # every rank by the single-GPU builder (``op.fit``: the same front end, the same refusals), and the FILTER is sharded: every rank
    # keeps its rows (``shard_of_graph``), as for ``sample_idx`` below.  (n_landmark: accepted, the filter never uses the operator.)
    replicated = [k for k in op.kwargs if k not in ("ksel", "sample_idx")]
    dense_kind = (op.thresh == 0 and op.decay is not None) or str(op.distance).lower().startswith("precomputed") \
        or str(op.distance).lower() in ("manhattan", "cityblock", "l1", "chebyshev") or min(int(op.knn), int(X.shape[0]) - 2) > 126
    if (replicated or dense_kind) and op.kwargs.get("sample_idx") is None:
        if not X.is_cuda:
            raise NotImplementedError("graph options {} on the row-sharded driver need the single-GPU builder on every rank (a GPU)".format(sorted(replicated)))
        op.fit(X)
        op.graph = shard_of_graph(op.graph, ops, comm)
        return op.transform(sample_labels)
    decay = float("inf") if op.decay is None else op.decay  # None: graphtools' unweighted kNN graph = a 0 / 1 kernel
    # the same front end as the single-GPU path (MELD._build_graph): reject NaN / infinity, and build the
    # graph on the PCA scores when n_pca < min(X.shape) (graphtools' Data._reduce_data; the reference's
    # default n_pca=100 triggers it on wide data).  Every rank holds all of X and computes the same
    # deterministic projection, so no communication is needed.
    if X.dim() != 2:
        raise ValueError("Expected a 2D data matrix, got shape {}".format(tuple(X.shape)))
    # (one pass: a NaN or an infinity anywhere makes its column sum non-finite; isfinite(X).all() is three)
    if not bool(torch.isfinite(X.sum(dim=0)).all()) and not bool(torch.isfinite(X).all()):
        raise ValueError("Input data contains NaN or infinity")
    op.data_nu = None
    if op.n_pca is not None and op.n_pca < min(tuple(X.shape)):
        from .pca import pca_project

        X = pca_project(X, op.n_pca, seed=42 if op.random_state is None else int(op.random_state)).contiguous()
        op.data_nu = X
    from .graph import metric_front_end

    # (cosine = the euclidean graph of the unit rows with the decay doubled: MELD._build_graph; every rank normalises all of X)
    op.X = X
    X, decay_m, bw_to_metric = metric_front_end(X, op.distance, op.decay)
    decay = float("inf") if decay_m is None else decay_m
    # (the label factorisation of transform starts under this rank's candidate search, as on one GPU)
    finish = op._prefactor_under_search(sample_labels, eligible=X.is_cuda) if hasattr(op, "_prefactor_under_search") else (lambda publish=True: None)
    try:
        if op.kwargs.get("sample_idx") is not None:
            # MNN graph (reference test/test_meld.py:31-40 forwards sample_idx to graphtools): its blocks between samples do not
            # follow the row shards, so every rank builds it whole -- the single-GPU builder -- and keeps its rows
            from .mnn import build_mnn_graph

            full = build_mnn_graph(X, op.kwargs["sample_idx"], knn=op.knn, decay=decay, thresh=op.thresh, anisotropy=op.anisotropy,
                                   ksel=op.kwargs.get("ksel"))
            op.graph = shard_of_graph(full, ops, comm)
        else:
            op.graph = build_sharded_graph(
                X, ops, comm, knn=op.knn, decay=decay, thresh=op.thresh, anisotropy=op.anisotropy, ksel=op.kwargs.get("ksel")
            )
    except BaseException:
        finish(publish=False)
        op._prefactored = None
        raise
    op.graph.bandwidth_to_metric = bw_to_metric
    finish()
    try:
        return op.transform(sample_labels)
    finally:
        op._prefactored = None
