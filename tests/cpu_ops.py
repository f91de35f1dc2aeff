"""NumPy/SciPy stand-in for ``meld_amd.graph.HipOps`` -- TEST INFRASTRUCTURE ONLY.

It lets the world_size-2 gloo tests drive the real communication code of
``meld_amd.distributed`` (sharding, all-to-all-v of transposed edges, all-gathers, all-reduces,
Lanczos / Chebyshev loops) on CPU tensors, where the HIP kernels cannot run.  The local
arithmetic is taken from the oracle; nothing here is imported by the product.
"""
import numpy as np
import torch
from scipy import sparse

from oracle import meld_oracle as mo


class CpuOps:
    name = "cpu-test"

    def __init__(self):
        self.device = torch.device("cpu")
        self._Kd = None

    def directed_kernel_coo(self, X, q_begin, q_count, knn, decay, thresh, ksel, tm=None, force_fallback=False):
        Xn = X.numpy()
        if self._Kd is None:  # every rank can afford the whole directed kernel at test sizes
            self._Kd, self._info = mo.knn_kernel(Xn, knn=knn, decay=None if np.isinf(decay) else decay, thresh=thresh, algorithm="brute",
                                                 return_intermediates=True)  # (decay = inf: graphtools' unweighted kNN graph, decay=None)
        K = self._Kd[q_begin : q_begin + q_count].tocoo()
        rows = K.row.astype(np.int64) + q_begin
        cols = K.col.astype(np.int64)
        keep = rows != cols  # the diagonal is carried analytically (K_ii = 1)
        rows, cols, v = rows[keep], cols[keep], 0.5 * K.data[keep]
        keys = np.concatenate([(rows << 32) | cols, (cols << 32) | rows])
        vals = np.concatenate([v, v])
        bw = torch.from_numpy(self._info["bandwidth"][q_begin : q_begin + q_count].copy())
        return torch.from_numpy(keys), torch.from_numpy(vals), bw, dict(ksel=ksel, n_flagged_rows=0, nnz_directed=len(v))

    def sort_pairs(self, keys, vals, N):
        order = torch.argsort(keys, stable=True)
        return keys[order].contiguous(), vals[order].contiguous()

    def partition_remote(self, keys, vals, rows_per_rank, world, rank, cap):
        """meld_coo_partition_remote: entries owed to the other ranks, [world, 2, cap] with sentinel keys, + counts."""
        k, v = keys.numpy(), vals.numpy()
        owner = np.minimum((k >> 32) // rows_per_rank, world - 1)
        send = np.full((world, 2, cap), -1, dtype=np.int64)
        counts = np.zeros(world, dtype=np.int32)
        for o in range(world):
            if o == rank:
                continue
            sel = np.nonzero(owner == o)[0]
            counts[o] = sel.shape[0]
            sel = sel[:cap]
            send[o, 0, : sel.shape[0]] = k[sel]
            send[o, 1, : sel.shape[0]] = v[sel].view(np.int64)
        return torch.from_numpy(send.reshape(-1)), torch.from_numpy(counts)

    def assemble_rows(self, keys, vals, row_begin, n_rows, N, foreign=False):
        if foreign:  # other ranks' rows and sentinel keys are ignored
            rows = keys >> 32
            own = (rows >= row_begin) & (rows < row_begin + n_rows)
            keys, vals = keys[own], vals[own]
        k = keys.numpy()
        uk, inv = np.unique(k, return_inverse=True)
        uv = np.zeros(uk.shape[0])
        np.add.at(uv, inv, vals.numpy())
        rows = (uk >> 32) - row_begin
        cols = (uk & 0xFFFFFFFF).astype(np.int32)
        rowptr = np.zeros(n_rows + 1, dtype=np.int64)
        np.add.at(rowptr, rows + 1, 1)
        rowptr = np.cumsum(rowptr)
        return torch.from_numpy(rowptr), torch.from_numpy(cols), torch.from_numpy(uv)

    def row_sums(self, rowptr, val, n_rows, diag):
        rp = rowptr.numpy()
        out = np.add.reduceat(np.concatenate([val.numpy(), [0.0]]), rp[:-1])[:n_rows]
        out[rp[1:] == rp[:-1]] = 0.0
        return torch.from_numpy(out + diag)

    def anisotropy(self, rowptr, col, val, n_rows, ksum_all, row_off, a):
        rp = rowptr.numpy()
        rows = np.repeat(np.arange(n_rows), np.diff(rp)[:n_rows])
        ks = ksum_all.numpy()
        v = val.numpy()
        v /= (ks[row_off + rows] * ks[col.numpy()]) ** a

    def anisotropy_degrees(self, rowptr, col, val, n_rows, ksum_all, row_off, a):
        self.anisotropy(rowptr, col, val, n_rows, ksum_all, row_off, a)
        return self.row_sums(rowptr, val, n_rows, 0.0)

    def dot_slots(self):
        return 4

    def cheby_step(self, G, p, x_full, x_row_off, z, y, r, alpha, beta, gamma, coef, dots=None):
        n = G.n_rows
        xf = x_full.numpy().reshape(x_full.shape[0], -1)
        W = sparse.csr_matrix((G.val.numpy(), G.col.numpy(), G.rowptr.numpy()[: n + 1]), shape=(n, xf.shape[0]))
        xl = xf[x_row_off : x_row_off + n]
        yv = alpha * (G.dw_dev.numpy()[:n, None] * xl - W @ xf) + beta * xl
        if gamma != 0.0:
            yv = yv + gamma * z.numpy().reshape(z.shape[0], -1)[:n]
        y.numpy().reshape(y.shape[0], -1)[:n] = yv
        if r is not None:
            r.numpy().reshape(r.shape[0], -1)[:n] += coef * yv
        if dots is not None:
            dots.zero_()
            s = self.dot_slots()
            dots[0] = float((yv * xl).sum())
            dots[s] = float((yv * yv).sum())

    def cheby_step_wide(self, G, p, x_full, x_row_off, z, y, alpha, beta, gamma):
        """meld_cheby_step_wide: the same step on a row-major [rows, p] signal (no accumulator, no dot products)."""
        self.cheby_step(G, p, x_full, x_row_off, z, y, None, alpha, beta, gamma, 0.0)

    # the four phases of the device-resident Lanczos iteration (meld_lanczos_spmv / _alpha / _axpy / _beta)
    def lanczos_spmv(self, G, x_full, z_local, y_local, state, dots):
        n = G.n_rows
        xf = x_full.numpy()
        W = sparse.csr_matrix((G.val.numpy(), G.col.numpy(), G.rowptr.numpy()[: n + 1]), shape=(n, xf.shape[0]))
        xl = xf[G.row_begin : G.row_begin + n]
        st = state.numpy()
        yv = st[3] * (G.dw_dev.numpy()[:n] * xl - W @ xf) + st[4] * z_local.numpy()[:n]
        y_local.numpy()[:n] = yv
        d = dots.numpy()
        d[0] += float(yv @ xl)  # the beta phase zeroed the slots
        d[self.dot_slots()] += float(yv @ yv)

    def lanczos_alpha(self, state, dots, nrm2, alphas, it):
        st = state.numpy()
        alpha = float(dots.numpy()[: self.dot_slots()].sum()) * st[0]
        alphas.numpy()[it] = alpha
        st[5] = -alpha * st[0]
        nrm2.zero_()

    def lanczos_axpy(self, x_local, y_local, state, nrm2):
        y = y_local.numpy()
        y += state.numpy()[5] * x_local.numpy()
        nrm2.numpy()[0] += float(y @ y)

    def lanczos_beta(self, state, nrm2, dots, betas, it):
        st = state.numpy()
        beta = float(np.sqrt(nrm2.numpy().sum()))
        betas.numpy()[it] = beta
        s_cur = st[0]
        st[1], st[2], st[0], st[3], st[4] = s_cur, beta, 1.0 / beta, 1.0 / beta, -beta * s_cur
        dots.zero_()

    # one-reduction form (meld_lanczos_fold / meld_lanczos_axpy3)
    def lanczos_fold(self, state, acc, alphas, betas, it):
        st, a = state.numpy(), acc.numpy()
        s = self.dot_slots()
        zu, uu = float(a[:s].sum()), float(a[2 * s :].sum())
        a[:] = 0.0
        n, alpha, n_prev = np.sqrt(uu), zu / uu, st[2]
        alphas.numpy()[it] = alpha
        if it > 0:
            betas.numpy()[it - 1] = n
        st[5], st[6], st[7], st[2] = 1.0 / n, -alpha / n, (-n / n_prev if it > 0 else 0.0), n

    def lanczos_axpy3(self, y_local, u_local, u_prev_local, state, nrm2):
        st, y = state.numpy(), y_local.numpy()
        y[:] = st[5] * y + st[6] * u_local.numpy() + st[7] * u_prev_local.numpy()
        nrm2.numpy()[0] += float(y @ y)

    def scale(self, x, a, r):
        r.copy_(a * x)

    def axpby(self, a, x, b, y, nrm2=None):
        y.copy_(a * x + b * y)
        if nrm2 is not None:
            nrm2.zero_()
            nrm2[0] = float((y * y).sum())
