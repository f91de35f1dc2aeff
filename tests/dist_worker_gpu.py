"""Worker of tests/test_gpu_parity.py::test_two_ranks_on_one_gpu: one rank of a 2-rank group in which
BOTH ranks drive the real HIP kernels (HipOps) on the same GPU.  RCCL refuses two ranks on one device,
so the collectives are staged through host memory over gloo -- the kernels, the shard arithmetic
(q_begin > 0, owner exchange, row offsets) and the order of the collectives are exactly the production
ones; only the transport differs."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path, n, d, knn, mode=""):
    dist.init_process_group("gloo")
    import meld_amd
    from meld_amd import distributed as mdist
    from oracle import meld_oracle as mo

    class StagedComm(mdist.Comm):
        """meld_amd.distributed.Comm with every collective staged through CPU tensors (gloo)."""

        def all_gather_rows(self, full, local):
            f = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(f, local.detach().cpu().contiguous(), group=self.group)
            full.copy_(f.to(full.device))

        def all_reduce_sum(self, t):
            c = t.detach().cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(c.to(t.device))
            return t

        def all_reduce_max(self, t):
            c = t.detach().cpu()
            dist.all_reduce(c, op=dist.ReduceOp.MAX, group=self.group)
            t.copy_(c.to(t.device))
            return t

        def exchange_by_owner(self, keys_sorted, vals_sorted, rows_per_rank):
            rk, rv = super().exchange_by_owner(keys_sorted.cpu(), vals_sorted.cpu(), rows_per_rank)
            return rk.to(keys_sorted.device), rv.to(vals_sorted.device)

        def exchange_fixed(self, send, cap):
            rk, rv = super().exchange_fixed(send.cpu(), cap)
            return rk.to(send.device), rv.to(send.device)

    torch.cuda.set_device(0)
    X, labels = mo.synthetic_cells(n, n_dims=d, seed=7)
    kw = {}
    if mode == "mnn":  # sample_idx: the MNN graph, built whole on every rank, rows sharded for the recurrences
        kw["sample_idx"] = np.random.default_rng(3).choice(["s0", "s1"], size=n)
    if mode.startswith("opt:"):  # graph keywords the sharded builder does not shard: replicated build, sharded filter
        import json

        kw.update(json.loads(mode[4:]))
    thresh = kw.pop("thresh", 1e-4)
    op = meld_amd.MELD(knn=knn, beta=40, chebyshev_order=25, verbose=0, decay=None if mode == "unweighted" else 40, thresh=thresh, **kw)
    dens = mdist.fit_transform_sharded(op, torch.from_numpy(X).cuda(), labels, comm=StagedComm())
    G = op.graph
    extra = {}
    if mode == "vfc":
        # BASELINE configs[4] on the sharded driver with the HIP kernels on every rank: the filter-bank
        # VertexFrequencyCluster fitted on this rank's shard (iterate all-gathered per SpMM, Gram matrices all-reduced)
        vfc = meld_amd.VertexFrequencyCluster(method="filterbank", n_probes=24, n_bands=6, window_sizes=np.array([1, 2, 4, 8]),
                                              chebyshev_order=48, random_state=3, n_clusters=3)
        vfc.fit(G)
        extra = dict(spec=vfc._fb_spectrogram.cpu().numpy(), ritz=vfc._fb["ritz"].cpu().numpy(), norm2=vfc._fb["window_norm2"].cpu().numpy())
    if mode in ("mnn", "unweighted", "vfc") or mode.startswith("opt:"):
        np.savez(out_path + ".rank{}".format(dist.get_rank()), dens=dens.values, lmax=G.lmax, row_begin=G.row_begin, n_rows=G.n_rows,
                 nnz_global=G.info["nnz_global"], **extra)
        dist.destroy_process_group()
        return
    # the ordering whose assignment passes were split over the ranks is the ordering one GPU computes alone
    from meld_amd.reorder import locality_permutation

    perm_equal = bool(torch.equal(G.perm, locality_permutation(op.X)))
    np.savez(
        out_path + ".rank{}".format(dist.get_rank()), dens=dens.values, lmax=G.lmax, row_begin=G.row_begin, n_rows=G.n_rows,
        nnz_global=G.info["nnz_global"], iters=G.lmax_info["iterations"], device_resident=bool(G.lmax_info.get("device_resident", False)),
        exchange=G.info["exchange"], perm_equal=perm_equal, all_reduces=G.lmax_info.get("all_reduces_per_iteration", 2),
    )
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] if len(sys.argv) > 5 else "")
