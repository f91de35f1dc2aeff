"""MELD graph filter on the device: the MI355X replacement of ``meld/filter.py``.

``filter(signal, graph, filter, beta, offset, order, solver, chebyshev_order)`` keeps the
reference signature (reference ``meld/filter.py:5-14``) and semantics:

* ``graph.estimate_lmax()`` first (``meld/filter.py:39``);
* spectral kernel ``heat``: exp(-beta |x/lmax - offset|^order), ``laplacian``:
  1/(1 + (beta |x/lmax - offset|)^order) (``meld/filter.py:42-50``), anything else raises
  ``NotImplementedError`` (``meld/filter.py:52-53``);
* ``solver="chebyshev"``: coefficients by the pygsp quadrature and the three-term recurrence
  [UPSTREAM pygsp ``compute_cheby_coeff`` / ``cheby_op``] -- the recurrence runs as one fused HIP
  kernel per order (``meld_cheby_step``) on the CSR weights resident in HBM.

Host code here only evaluates the M+1 scalar coefficients and sequences kernel launches.
"""
from __future__ import annotations

from ._options import is_set, opt

import os

import numpy as np
import torch

from ._lib import check, get_lib, ptr

__all__ = ["filter", "filter_sweep", "chebyshev_coefficients", "spectral_kernel", "lanczos_lmax", "chebyshev_apply", "IndicatorSignal"]


class IndicatorSignal:
    """A scaled one-hot signal given by its label codes: column ``codes[i]`` of row ``i`` holds
    ``scale[codes[i]]`` (1 when ``scale`` is None), everything else is 0.  ``MELD.transform``
    passes this instead of the dense [N, p] matrix so that only the codes cross PCIe."""

    def __init__(self, codes, n_columns, scale=None):
        # codes: host int array, or an int64 device tensor (labels factorised on the device)
        self.codes = codes if isinstance(codes, torch.Tensor) else np.ascontiguousarray(codes, dtype=np.int64)
        self.n_columns = int(n_columns)
        self.scale = None if scale is None else np.asarray(scale, dtype=np.float64)
        self.shape = (int(self.codes.shape[0]), self.n_columns)

    def host_codes(self):
        return self.codes.cpu().numpy() if isinstance(self.codes, torch.Tensor) else self.codes

    def to_dense(self):
        codes = self.host_codes()
        out = np.zeros(self.shape, dtype=np.float64)
        out[np.arange(self.shape[0]), codes] = 1.0 if self.scale is None else self.scale[codes]
        return out

    def to_device_ordered(self, device, perm, n_pad):
        """The [n_pad, p] signal on the device with row i = the indicator row of cell perm[i] (perm None: i) and zero rows
        behind the N cells: ``meld_indicator_signal``, one pass instead of zeros / gather / scatter / index_select / cat."""
        if isinstance(self.codes, torch.Tensor):
            codes = self.codes.to(device=device, dtype=torch.int64)
        else:
            codes = torch.from_numpy(self.codes.astype(np.int32)).to(device).to(torch.int64)
        scale = None if self.scale is None else torch.from_numpy(self.scale).to(device)
        out = torch.empty((int(n_pad), self.n_columns), dtype=torch.float64, device=device)
        check(get_lib().meld_indicator_signal(ptr(codes.contiguous()), ptr(scale), ptr(perm), self.shape[0], int(n_pad), self.n_columns,
                                              ptr(out), torch.cuda.current_stream().cuda_stream), "meld_indicator_signal")
        return out

    def to_device(self, device):
        if isinstance(self.codes, torch.Tensor):
            codes = self.codes.to(device=device, dtype=torch.int64)
        else:
            codes = torch.from_numpy(self.codes.astype(np.int32)).to(device).to(torch.int64)
        out = torch.zeros(self.shape, dtype=torch.float64, device=device)
        if self.scale is None:
            vals = torch.ones(self.shape[0], dtype=torch.float64, device=device)
        else:
            vals = torch.from_numpy(self.scale).to(device)[codes]
        out[torch.arange(self.shape[0], device=device), codes] = vals
        return out


def _stream():
    return torch.cuda.current_stream().cuda_stream


_PINNED = {}  # (shape, dtype) -> pinned host staging buffer of the last result copied back


class _PinnedPool:
    """Pinned result buffers handed to the caller WITHOUT a host copy.  The densities leave the device through pinned memory
    (pageable copies run at a few GB/s) and used to be copied once more into a fresh array because the staging buffer is
    reused by the next call (0.4 ms for 16 MB).  Now the caller's array IS the pinned buffer: a finalizer on that array --
    which fires when the array and every view of it (the DataFrame's) are gone -- returns the buffer to the pool.  At most
    ``MAX_OUT`` buffers of a shape are lent out at a time (pinning a new one costs milliseconds and pinned memory is a
    shared resource); beyond that the result is copied as before."""

    MAX_OUT = 4
    MAX_FREE_BYTES = 256 << 20  # page-locked memory parked in the pool (not lent out) at most

    def __init__(self):
        self.free = {}  # key -> [tensor, ...]
        self.out = {}   # key -> number lent out
        self.last_path = None  # "pinned" / "copy": which way the last result went (observable: tests, diagnostics)

    def _free_bytes(self):
        return sum(t.numel() * t.element_size() for lst in self.free.values() for t in lst)

    def _evict(self, keep_key):
        """Page-locked memory is not swappable: buffers of shapes other than the one in use are dropped first (a process that
        runs differently sized inputs -- subsampling sweeps, parameter searches -- would otherwise keep MAX_OUT buffers of
        every shape it ever saw), then surplus buffers of this shape, until the parked total is under MAX_FREE_BYTES."""
        for key in [k for k in self.free if k != keep_key]:
            if self._free_bytes() <= self.MAX_FREE_BYTES:
                return
            self.free.pop(key, None)
        lst = self.free.get(keep_key, [])
        while len(lst) > 1 and self._free_bytes() > self.MAX_FREE_BYTES:
            lst.pop()

    def lend(self, r):
        """r: device tensor -> numpy array with its values (device synchronised), or None when the pool is exhausted."""
        import weakref

        key = (tuple(r.shape), r.dtype)
        free = self.free.setdefault(key, [])
        if not free and self.out.get(key, 0) >= self.MAX_OUT:
            self.last_path = "copy"
            return None
        if not free:
            self._evict(key)  # (about to pin a new buffer: make room first)
        self.last_path = "pinned"
        stage = free.pop() if free else torch.empty(r.shape, dtype=r.dtype, pin_memory=True)
        stage.copy_(r, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        arr = stage.numpy()
        self.out[key] = self.out.get(key, 0) + 1

        def back(pool=self, key=key, stage=stage):
            pool.out[key] = pool.out.get(key, 1) - 1
            pool.free.setdefault(key, []).append(stage)
            pool._evict(key)

        weakref.finalize(arr, back)
        return arr


_POOL = _PinnedPool()


def spectral_kernel(name, beta, offset, order, lmax):
    """h(lambda) of reference ``meld/filter.py:42-53``."""
    lname = name.lower()
    if lname == "laplacian":
        return lambda x: 1 / (1 + (beta * np.abs(x / lmax - offset)) ** order)
    elif lname == "heat":
        return lambda x: np.exp(-beta * np.abs(x / lmax - offset) ** order)
    raise NotImplementedError


def chebyshev_coefficients(h, lmax, m):
    """c_0..c_m for h on [0, lmax] with m+1 quadrature points
    [UPSTREAM pygsp 0.5.1 ``compute_cheby_coeff(f, m)``; N = m + 1]."""
    m = int(m)
    n = m + 1
    a = lmax / 2.0
    k = np.arange(n)
    nodes = np.cos(np.pi * (k + 0.5) / n)
    hv = h(a * nodes + a)
    c = np.empty(m + 1)
    for o in range(m + 1):
        c[o] = 2.0 / n * np.dot(hv, np.cos(np.pi * o * (k + 0.5) / n))
    return c


def _ops_of(G):
    ops = getattr(G, "ops", None)
    if ops is None:
        from .graph import HipOps

        ops = HipOps(G.val.device)
        G.ops = ops
    return ops


def _local(G, full):
    """View of the local rows of a full-length [N_pad, ...] buffer."""
    return full[G.row_begin : G.row_begin + G.rows_pad]


def chebyshev_apply(G, signal, coeffs, lmax):
    """r = sum_k c_k T_k(2 L / lmax - I) signal, on the device.

    ``signal``: fp64 tensor [N_pad, p] on the graph's device (the full signal; every rank of a
    sharded graph holds it).  Returns the local rows of r ([rows_pad, p]; the whole result on a
    single GPU).

    T0 = s; T1 = (L s - a2 s)/a1; r = c0/2 T0 + c1 T1; Tk = (2/a1)(L - a2 I) T(k-1) - T(k-2);
    r += ck Tk  [UPSTREAM pygsp ``cheby_op``], a1 = a2 = lmax/2.  Two full-length ping-pong
    buffers: each step overwrites the local rows of T(k-2) with T(k); on a sharded graph the
    local slices are then all-gathered (one collective per step)."""
    ops = _ops_of(G)
    comm = getattr(G, "comm", None)
    c = np.asarray(coeffs, dtype=np.float64)
    if c.ndim == 2:
        return _chebyshev_apply_batch(G, signal, c, lmax)
    if c.shape[0] < 2:
        raise TypeError("The coefficients have an invalid shape")
    t_old = signal.contiguous().clone()
    if t_old.shape[0] != G.n_pad:
        raise ValueError("signal has {} rows, the graph expects {}".format(t_old.shape[0], G.n_pad))
    p = int(t_old.shape[1])
    a1 = a2 = float(lmax) / 2.0
    t_cur = torch.zeros_like(t_old)
    r = torch.empty_like(_local(G, t_old))
    ops.scale(_local(G, t_old), 0.5 * c[0], r)
    from .graph import _EventSpan

    ops.cheby_step(G, p, t_old, G.row_begin, None, _local(G, t_cur), r, 1.0 / a1, -a2 / a1, 0.0, c[1])
    if comm is not None:
        comm.all_gather_rows(t_cur, _local(G, t_cur))
    with _EventSpan("cheby_steps", steps=int(c.shape[0] - 2), N=G.N, p=p, nnz=G.nnz):
        if comm is None and c.shape[0] > 2 and hasattr(ops, "cheby_run") and opt("MELD_CHEBY_RUN", "1") != "0" \
                and ops.cheby_run(G, p, t_old, t_cur, r, c, 2.0 / a1, -2.0 * a2 / a1):
            return r  # (one call for all the steps; r is read and written every other step only)
        if comm is not None and c.shape[0] > 2 and hasattr(ops, "cheby_run_sharded") and opt("MELD_CHEBY_RUN", "1") != "0" \
                and ops.cheby_run_sharded(G, p, t_old, t_cur, r, c, 2.0 / a1, -2.0 * a2 / a1) is not None:
            return r  # (row shard on RCCL: kernel + all-gather of every step enqueued from one C call)
        for k in range(2, c.shape[0]):
            # T_k overwrites the local rows of T_{k-2} (z and y alias; read-before-write per element)
            loc = _local(G, t_old)
            ops.cheby_step(G, p, t_cur, G.row_begin, loc, loc, r, 2.0 / a1, -2.0 * a2 / a1, -1.0, c[k])
            if comm is not None:
                comm.all_gather_rows(t_old, loc)
            t_old, t_cur = t_cur, t_old
    return r


def _chebyshev_apply_batch(G, signal, C, lmax):
    """B filters of the same signal in ONE pass over the graph (the
    parameter-sweep mode of reference ``meld/benchmark.py:186-200``): the polynomials T_k(L) s do not depend on
    the filter, only the coefficients c[b, k] do, so the recurrence runs once and every step is added to all
    B accumulators (one fused elementwise pass over [B, rows, p]).  Returns [B, rows_pad, p]."""
    ops = _ops_of(G)
    comm = getattr(G, "comm", None)
    if C.shape[1] < 2:
        raise TypeError("The coefficients have an invalid shape")
    B = C.shape[0]
    dev = signal.device
    Cd = torch.from_numpy(np.ascontiguousarray(C)).to(dev)
    t_old = signal.contiguous().clone()
    if t_old.shape[0] != G.n_pad:
        raise ValueError("signal has {} rows, the graph expects {}".format(t_old.shape[0], G.n_pad))
    p = int(t_old.shape[1])
    a1 = a2 = float(lmax) / 2.0
    t_cur = torch.zeros_like(t_old)
    loc0 = _local(G, t_old)
    R = (0.5 * Cd[:, 0]).view(B, 1, 1) * loc0.unsqueeze(0)
    ops.cheby_step(G, p, t_old, G.row_begin, None, _local(G, t_cur), None, 1.0 / a1, -a2 / a1, 0.0, 0.0)
    R.addcmul_(_local(G, t_cur).unsqueeze(0).expand_as(R), Cd[:, 1].view(B, 1, 1).expand_as(R))
    if comm is not None:
        comm.all_gather_rows(t_cur, _local(G, t_cur))
    for k in range(2, C.shape[1]):
        loc = _local(G, t_old)
        ops.cheby_step(G, p, t_cur, G.row_begin, loc, loc, None, 2.0 / a1, -2.0 * a2 / a1, -1.0, 0.0)
        R.addcmul_(loc.unsqueeze(0).expand_as(R), Cd[:, k].view(B, 1, 1).expand_as(R))
        if comm is not None:
            comm.all_gather_rows(t_old, loc)
        t_old, t_cur = t_cur, t_old
    return R


def _ritz_check(alphas, betas, tol):
    """(theta, relative residual, breakdown) of the k x k Lanczos tridiagonal; betas[k-1] is the
    residual norm of the last step."""
    k = len(alphas)
    if k == 1:
        return float(alphas[0]), abs(betas[0]) / max(abs(float(alphas[0])), 1e-300)
    # the tridiagonal solver, top pair only (a dense eigh of the k x k matrix goes through the threaded BLAS once
    # k passes ~64: on a 256-core host that cost tens of milliseconds per check -- 55 ms per fit at 500k cells,
    # where the recurrence needs 100 iterations)
    from scipy.linalg import eigh_tridiagonal

    ev, evec = eigh_tridiagonal(np.asarray(alphas, dtype=np.float64), np.asarray(betas[: k - 1], dtype=np.float64),
                                select="i", select_range=(k - 1, k - 1))
    theta = float(ev[0])
    resid = abs(betas[k - 1] * evec[-1, 0]) / max(abs(theta), 1e-300)
    return theta, resid


class _LanczosHostSide:
    """What the host's half of the device-resident Lanczos loop keeps per device: a side stream that carries the tridiagonal
    entries to a pinned buffer behind every batch of iterations, so that the main stream never waits for a convergence check."""

    _local = None  # threading.local: one (side stream, pinned buffer) per thread and device, released when the thread ends

    def __init__(self, dev, max_iter):
        self.side = torch.cuda.Stream(device=dev)
        self.ab = torch.empty(2, max_iter, dtype=torch.float64, pin_memory=True)

    @classmethod
    def of(cls, dev, max_iter):
        import threading

        # (per thread: two estimates running side by side must not share the pinned buffer; thread-local storage rather than a
        # table keyed by thread id -- ids are recycled and a table is never pruned)
        if cls._local is None:
            cls._local = threading.local()
        table = cls._local.__dict__.setdefault("per_device", {})
        key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
        h = table.get(key)
        if h is None or h.ab.shape[1] < max_iter:
            h = table[key] = cls(dev, max_iter)
        return h


def _lanczos_lmax_device(G, ops, u0, tol, max_iter, check_every):
    """Single-GPU Lanczos with every scalar on the device (``meld_lanczos_steps``): iterations are enqueued in batches -- a
    first batch of 4 checks' worth, then one check per ``check_every`` iterations -- and the tridiagonal entries of a batch
    travel to the host on a side stream while the NEXT batch already runs: the convergence check (a k x k tridiagonal
    eigenproblem on the host) overlaps with device work instead of idling the GPU once per batch (12 idle gaps of ~0.1 ms at
    500k cells, 75 iterations).  Once the check has passed, a flag set from the side stream voids what is left of the batch in
    flight (its launches return at once, ``stop`` of ``meld_pt_lanczos_steps``), so the overlap costs an iteration or two of
    wasted work, not a batch (``MELD_LANCZOS_SPECULATE=0``: the serial loop)."""
    dev, n = G.val.device, G.N
    slots = ops.dot_slots()
    V = torch.zeros(3, n, dtype=torch.float64, device=dev)
    V[1].copy_(u0[:n])
    state = torch.zeros(8, dtype=torch.float64, device=dev)
    inv = 1.0 / torch.linalg.vector_norm(V[1])
    state[0] = inv
    state[3] = inv
    ab_d = torch.zeros(2, max_iter, dtype=torch.float64, device=dev)
    alphas_d, betas_d = ab_d[0], ab_d[1]
    scratch = torch.zeros(8 * slots, dtype=torch.float64, device=dev)  # (the tiled loop keeps parity buffers there: 8 x slots)
    host = _LanczosHostSide.of(dev, max_iter)
    main = torch.cuda.current_stream(dev)
    ab_d.record_stream(host.side)
    in_flight = 2 if opt("MELD_LANCZOS_SPECULATE", "1") != "0" else 1
    stop = torch.zeros(1, dtype=torch.int32, device=dev)  # set once the check has passed: what is left of the batch in flight is void
    stop.record_stream(host.side)

    def finish(theta, info):
        if in_flight > 1:
            with torch.cuda.stream(host.side):
                stop.fill_(1)
        return theta, info

    def enqueue(lo, hi):
        """iterations [lo, hi) on the main stream, their alphas / betas to the pinned buffer behind them on the side stream"""
        ops.lanczos_steps(G, V, state, alphas_d, betas_d, lo, hi - lo, scratch, stop)
        ran = torch.cuda.Event()
        ran.record(main)
        host.side.wait_event(ran)
        with torch.cuda.stream(host.side):
            host.ab[0, lo:hi].copy_(alphas_d[lo:hi], non_blocking=True)
            host.ab[1, lo:hi].copy_(betas_d[lo:hi], non_blocking=True)
            landed = torch.cuda.Event()
            landed.record(host.side)
        return lo, hi, landed

    it = min(4 * check_every, max_iter)
    pending = [enqueue(0, it)]
    theta, resid, examined = 0.0, float("inf"), 0
    while pending or it < max_iter:
        while it < max_iter and len(pending) < in_flight:
            hi = min(it + check_every, max_iter)
            pending.append(enqueue(it, hi))
            it = hi
        lo, hi, landed = pending.pop(0)
        landed.synchronize()
        alphas, betas = host.ab[0, :hi].numpy(), host.ab[1, :hi].numpy()
        # examine the prefixes a per-iteration loop would have examined
        for k in range(lo + 1, hi + 1):
            done = betas[k - 1] <= 1e-14 * max(abs(alphas[k - 1]), 1e-300) or not np.isfinite(betas[k - 1])
            if k % check_every == 0 or done or k == max_iter:
                theta, resid = _ritz_check(alphas[:k], betas[:k], tol)
                if resid <= tol or done:
                    return finish(theta, dict(iterations=k, residual=resid, tol=tol, device_resident=True, enqueued=it))
        examined = hi
    return theta, dict(iterations=examined, residual=resid, tol=tol, device_resident=True, enqueued=it)


def _lanczos_lmax_phases(G, ops, comm, u0, tol, max_iter, check_every):
    """Device-resident Lanczos on a row-sharded graph: the four phases of an iteration
    (``meld_lanczos_spmv / _alpha / _axpy / _beta``) interleaved with the all-reduces of the partial
    sums and the all-gather of the new vector, all stream-ordered -- one host synchronisation per batch
    of iterations instead of two per iteration.  Every rank reads back the same alphas / betas, so all
    ranks take the same decisions and issue the same collectives."""
    dev, n_pad = G.val.device, G.n_pad
    slots = ops.dot_slots()
    V = torch.zeros(3, n_pad, dtype=torch.float64, device=dev)
    V[1].copy_(u0)
    state = torch.zeros(8, dtype=torch.float64, device=dev)
    inv = 1.0 / torch.linalg.vector_norm(V[1])
    state[0] = inv
    state[3] = inv
    alphas_d = torch.zeros(max_iter, dtype=torch.float64, device=dev)
    betas_d = torch.zeros(max_iter, dtype=torch.float64, device=dev)
    dots = torch.zeros(2 * slots, dtype=torch.float64, device=dev)
    nrm2 = torch.zeros(slots, dtype=torch.float64, device=dev)
    it, theta, resid = 0, 0.0, float("inf")
    batch = 4 * check_every
    while it < max_iter:
        n_iter = min(batch, max_iter - it)
        for k in range(it, it + n_iter):
            u_prev, u, y = V[k % 3], V[(k + 1) % 3], V[(k + 2) % 3]
            ops.lanczos_spmv(G, u, _local(G, u_prev), _local(G, y), state, dots)
            if comm is not None:
                comm.all_reduce_sum(dots)
            ops.lanczos_alpha(state, dots, nrm2, alphas_d, k)
            ops.lanczos_axpy(_local(G, u), _local(G, y), state, nrm2)
            if comm is not None:
                comm.all_reduce_sum(nrm2)
            ops.lanczos_beta(state, nrm2, dots, betas_d, k)
            if comm is not None:
                comm.all_gather_rows(y, _local(G, y))
        it_new = it + n_iter
        ab = torch.stack([alphas_d[:it_new], betas_d[:it_new]]).cpu().numpy()  # the one synchronisation per batch
        alphas, betas = ab[0], ab[1]
        for k in range(it + 1, it_new + 1):
            done = betas[k - 1] <= 1e-14 * max(abs(alphas[k - 1]), 1e-300) or not np.isfinite(betas[k - 1])
            if k % check_every == 0 or done or k == max_iter:
                theta, resid = _ritz_check(alphas[:k], betas[:k], tol)
                if resid <= tol or done:
                    return theta, dict(iterations=k, residual=resid, tol=tol, device_resident=True)
        it = it_new
        batch = check_every
    return theta, dict(iterations=it, residual=resid, tol=tol, device_resident=True)


def _lanczos_lmax_folded(G, ops, comm, u0, tol, max_iter, check_every):
    """The sharded iteration with ONE all-reduce: the iterate stays un-normalised (u_{k+1} = w_k), so the SpMV
    z = L u_k needs no scalar and the sums <z, u_k> (SpMV) and |u_k|^2 (the axpy that formed u_k) travel in one
    buffer (``meld_lanczos_fold`` / ``meld_lanczos_axpy3``).  Per iteration: SpMV, all-reduce of 3 x slots doubles,
    a one-wave scalar kernel, the three-term update, all-gather of the new vector -- against two all-reduces in
    ``_lanczos_lmax_phases``.  beta_k = |u_{k+1}| is known one iteration late, so a batch runs one iteration past
    the prefix it examines; the prefixes examined and the value returned are those of the other two drivers."""
    dev, n_pad = G.val.device, G.n_pad
    slots = ops.dot_slots()
    V = torch.zeros(3, n_pad, dtype=torch.float64, device=dev)
    V[1].copy_(u0)
    state = torch.zeros(8, dtype=torch.float64, device=dev)
    state[3] = 1.0  # SpMV arguments: z = 1 * L u + 0 * u_prev
    alphas_d = torch.zeros(max_iter + 1, dtype=torch.float64, device=dev)
    betas_d = torch.zeros(max_iter + 1, dtype=torch.float64, device=dev)
    acc = torch.zeros(3 * slots, dtype=torch.float64, device=dev)
    nrm2 = acc[2 * slots :]
    u_loc0 = _local(G, V[1])
    nrm2[0] = torch.dot(u_loc0, u_loc0)
    it, theta, resid = 0, 0.0, float("inf")  # it = iterations run; prefixes 1 .. it - 1 have their beta
    examined = 0
    target = min(4 * check_every, max_iter)
    while examined < max_iter:
        n_batch = min(target + 1, max_iter + 1) - it
        if comm is not None and n_batch > 0 and hasattr(ops, "lanczos_steps_sharded") \
                and ops.lanczos_steps_sharded(G, V, state, acc, alphas_d, betas_d, it, n_batch):
            it += n_batch  # (row shard on RCCL: the whole batch enqueued from one C call)
        while it < min(target + 1, max_iter + 1):
            k = it
            u_prev, u, y = V[k % 3], V[(k + 1) % 3], V[(k + 2) % 3]
            ops.lanczos_spmv(G, u, _local(G, u_prev), _local(G, y), state, acc)
            if comm is not None:
                comm.all_reduce_sum(acc)
            ops.lanczos_fold(state, acc, alphas_d, betas_d, k)
            ops.lanczos_axpy3(_local(G, y), _local(G, u), _local(G, u_prev), state, nrm2)
            if comm is not None:
                comm.all_gather_rows(y, _local(G, y))
            it += 1
        ab = torch.stack([alphas_d[:it], betas_d[:it]]).cpu().numpy()  # the one synchronisation per batch
        alphas, betas = ab[0], ab[1]
        for k in range(examined + 1, it):  # beta_k = betas[k - 1] was written by iteration k (0-based), i.e. k + 1 <= it
            done = betas[k - 1] <= 1e-14 * max(abs(alphas[k - 1]), 1e-300) or not np.isfinite(betas[k - 1])
            if k % check_every == 0 or done or k == max_iter:
                theta, resid = _ritz_check(alphas[:k], betas[:k], tol)
                if resid <= tol or done:
                    return theta, dict(iterations=k, residual=resid, tol=tol, device_resident=True, all_reduces_per_iteration=1)
        examined = it - 1
        target = min(examined + check_every, max_iter)
    return theta, dict(iterations=examined, residual=resid, tol=tol, device_resident=True, all_reduces_per_iteration=1)


def lanczos_lmax(G, tol=3e-4, max_iter=300, check_every=5, seed=0):
    """Largest eigenvalue of L = diag(dw) - W by the Lanczos recurrence on the device SpMV.

    Vectors stay un-normalised on the device (u_k = beta_{k-1} v_k); the 1/beta scalings are folded
    into the alpha/gamma arguments of ``meld_cheby_step``, which also returns <y, u>, so one
    iteration = one SpMV kernel + one axpby kernel (which returns |w|^2 directly -- the shortcut
    |y|^2 - alpha^2 is unstable) + two small read-backs (+ two scalar all-reduces and one
    all-gather of the new vector on a sharded graph).  Convergence: relative Ritz residual
    |beta_m s_m| / theta <= tol (s = last component of the top eigenvector of the tridiagonal
    matrix); the eigenvalue error is then ~ tol^2 / gap, far below tol (measured < 1e-8 relative
    at tol = 1e-4, against the 1e-4..1e-5 run-to-run spread of the reference's own estimate)."""
    ops = _ops_of(G)
    comm = getattr(G, "comm", None)
    dev = G.val.device
    slots = ops.dot_slots()
    # deterministic start vector, generated where it is used (a 1M-entry CPU randn costs 15 ms)
    idx = torch.arange(G.n_pad, dtype=torch.float64, device=dev)
    u = torch.frac(torch.sin(idx * 12.9898 + float(seed) + 1.0) * 43758.5453) - 0.5
    u[G.N :] = 0.0
    max_iter = min(max_iter, G.N)
    if comm is None and hasattr(ops, "lanczos_steps") and G.n_pad == G.N:
        return _lanczos_lmax_device(G, ops, u, tol, max_iter, check_every)
    if hasattr(ops, "lanczos_fold") and opt("MELD_LANCZOS_FOLD", "1") != "0":
        return _lanczos_lmax_folded(G, ops, comm, u, tol, max_iter, check_every)
    if hasattr(ops, "lanczos_spmv"):
        return _lanczos_lmax_phases(G, ops, comm, u, tol, max_iter, check_every)
    nrm = float(torch.linalg.vector_norm(u).item())
    u_prev = torch.zeros(G.n_pad, dtype=torch.float64, device=dev)
    y = torch.zeros(G.n_pad, dtype=torch.float64, device=dev)
    dots = torch.zeros(2 * slots, dtype=torch.float64, device=dev)
    nrm2 = torch.zeros(slots, dtype=torch.float64, device=dev)

    def total(t):
        s = t.sum().reshape(1)
        if comm is not None:
            comm.all_reduce_sum(s)
        return float(s.item())

    alphas, betas = [], []
    s_cur = 1.0 / nrm  # v_k = s_cur * u
    s_prev = 0.0
    beta_prev = 0.0
    theta, resid = 0.0, float("inf")
    it = 0
    max_iter = min(max_iter, G.N)
    while it < max_iter:
        # y = L v_k - beta_{k-1} v_{k-1}   (local rows)
        ops.cheby_step(G, 1, u, G.row_begin, _local(G, u_prev), _local(G, y), None, s_cur, 0.0, -beta_prev * s_prev, 0.0, dots)
        alpha = total(dots[:slots]) * s_cur  # <y, v_k>
        # w = y - alpha v_k (stored in y), beta = |w|
        ops.axpby(-alpha * s_cur, _local(G, u), 1.0, _local(G, y), nrm2)
        beta = float(np.sqrt(total(nrm2)))
        alphas.append(alpha)
        it += 1
        done = beta <= 1e-14 * max(abs(alpha), 1e-300)
        if it % check_every == 0 or done or it == max_iter:
            theta, resid = _ritz_check(alphas, betas + [beta], tol)  # (the tridiagonal solver: see there for why not a dense eigh)
            if resid <= tol or done:
                break
        if comm is not None:
            comm.all_gather_rows(y, _local(G, y))
        betas.append(beta)
        u_prev, u, y = u, y, u_prev
        s_prev, s_cur = s_cur, 1.0 / beta
        beta_prev = beta
    return theta, dict(iterations=it, residual=resid, tol=tol)


def filter_sweep(signal, graph, filter, betas, offset=0, order=1, chebyshev_order=None):  # noqa: A002
    """``[filter(signal, graph, filter, beta, ...) for beta in betas]`` in one pass over the graph
    (parameter-sweep mode; the reference's benchmark loop re-runs the whole
    filter per beta, ``meld/benchmark.py:186-200``).  Returns an ndarray [len(betas), N, p]."""
    graph.estimate_lmax()
    betas = [float(b) for b in betas]
    if not betas:
        raise ValueError("betas must not be empty")
    if chebyshev_order is None:
        chebyshev_order = 30
    C = np.stack([chebyshev_coefficients(spectral_kernel(filter, b, offset, order, graph.lmax), graph.lmax, chebyshev_order) for b in betas])
    is_ind = isinstance(signal, IndicatorSignal)
    sig = signal if is_ind else np.asarray(getattr(signal, "values", signal), dtype=np.float64)
    if sig.shape[0] != graph.N:
        raise ValueError("First dimension should be the number of nodes G.N = {}, got {}.".format(graph.N, sig.shape))
    if not is_ind and sig.ndim == 1:
        sig = sig[:, None]
    dev = graph.val.device
    s_dev = sig.to_device(dev) if is_ind else torch.from_numpy(np.ascontiguousarray(sig)).to(dev)
    perm = getattr(graph, "perm", None)
    if perm is not None:
        s_dev = s_dev.index_select(0, perm)
    if graph.n_pad != graph.N:
        s_dev = torch.cat([s_dev, torch.zeros(graph.n_pad - graph.N, s_dev.shape[1], dtype=s_dev.dtype, device=dev)])
    R = chebyshev_apply(graph, s_dev, C, graph.lmax)  # [B, rows_pad, p]
    comm = getattr(graph, "comm", None)
    if comm is not None:
        full = torch.empty(R.shape[0], graph.n_pad, R.shape[2], dtype=R.dtype, device=dev)
        for b in range(R.shape[0]):
            comm.all_gather_rows(full[b], R[b].contiguous())
        R = full
    R = R[:, : graph.N]
    if perm is not None:
        out = torch.empty_like(R)
        out[:, perm] = R
        R = out
    return R.cpu().numpy()


def filter(signal, graph, filter, beta, offset=0, order=1, solver="chebyshev", chebyshev_order=None):  # noqa: A001,A002
    """Implements the MELD filter for sample-associated density estimation (reference
    ``meld/filter.py:5-61``).  ``signal`` is array-like [N] or [N, p] (DataFrame accepted); returns
    an ``ndarray`` squeezed like pygsp's ``Filter.filter`` output."""
    graph.estimate_lmax()
    h = spectral_kernel(filter, beta, offset, order, graph.lmax)  # raises NotImplementedError

    is_ind = isinstance(signal, IndicatorSignal)
    sig = signal if is_ind else np.asarray(getattr(signal, "values", signal), dtype=np.float64)
    if sig.shape[0] != graph.N:
        raise ValueError("First dimension should be the number of nodes G.N = {}, got {}.".format(graph.N, sig.shape))
    if not is_ind:
        if sig.ndim == 1:
            sig = sig[:, None]
        if sig.ndim != 2:
            raise ValueError("At most 2 dimensions are supported.")
    dev = graph.val.device

    if solver == "chebyshev":
        if chebyshev_order is None:
            chebyshev_order = 30  # pygsp's default order
        c = chebyshev_coefficients(h, graph.lmax, chebyshev_order)
        perm = getattr(graph, "perm", None)
        if is_ind and dev.type == "cuda":
            # the scaled one-hot straight in the device's row order, padding rows included (meld_indicator_signal: one pass)
            s_dev = sig.to_device_ordered(dev, perm, graph.n_pad)
        else:
            s_dev = sig.to_device(dev) if is_ind else torch.from_numpy(np.ascontiguousarray(sig)).to(dev)
            if perm is not None:  # device arrays live in the locality order
                s_dev = s_dev.index_select(0, perm)
            if graph.n_pad != graph.N:  # sharded graph: isolated padding rows at the end
                s_dev = torch.cat([s_dev, torch.zeros(graph.n_pad - graph.N, s_dev.shape[1], dtype=s_dev.dtype, device=dev)])
        r = chebyshev_apply(graph, s_dev, c, graph.lmax)
        comm = getattr(graph, "comm", None)
        if comm is not None:
            r_full = torch.empty_like(s_dev)
            comm.all_gather_rows(r_full, r)
            r = r_full
        r = r[: graph.N]
        if perm is not None:
            r_orig = torch.empty_like(r)
            if r.is_cuda and r.dtype == torch.float64 and r.is_contiguous():
                check(get_lib().meld_scatter_rows_f64(ptr(r), ptr(perm), int(r.shape[0]), int(r.shape[1]), ptr(r_orig), _stream()), "meld_scatter_rows_f64")
            else:
                r_orig[perm] = r
            r = r_orig
        # D2H through a pinned staging buffer kept on the graph (pageable copies run at a few GB/s)
        out = _POOL.lend(r) if (r.is_cuda and opt("MELD_PINNED_RESULT", "1") != "0") else None
        if out is not None:
            pass
        elif r.is_cuda:
            stage = _PINNED.get((tuple(r.shape), r.dtype))  # (process-wide: a new graph per fit would pin 16 MB each time)
            if stage is None:
                _PINNED.clear()
                stage = torch.empty(r.shape, dtype=r.dtype, pin_memory=True)
                _PINNED[(tuple(r.shape), r.dtype)] = stage
            stage.copy_(r, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            out = stage.numpy().copy()
        else:  # CPU tensors only occur in the gloo tests of the sharded driver
            out = r.numpy().copy()
    elif solver == "exact":
        from .dense import exact_filter

        out = exact_filter(graph, sig.to_dense() if is_ind else sig,
                           lambda lm: spectral_kernel(filter, beta, offset, order, lm))
    else:
        raise ValueError("Unknown method {}.".format(solver))
    return out.squeeze()
