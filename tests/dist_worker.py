"""Worker of tests/test_distributed_gloo.py: one rank of a gloo group running the sharded
fit_transform of meld_amd.distributed on CPU tensors with the NumPy stand-in ops."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path, n, d, knn, n_labels, n_pca=0):
    dist.init_process_group("gloo")
    import meld_amd
    from meld_amd import distributed as mdist
    from oracle import meld_oracle as mo
    from tests.cpu_ops import CpuOps

    X, labels = mo.synthetic_cells(n, n_dims=d, seed=7)
    if n_labels == 3:
        labels = np.random.default_rng(1).choice(["A", "B", "C"], size=n)
    mode = os.environ.get("MELD_TEST_MODE", "")
    op = meld_amd.MELD(knn=knn, beta=40, chebyshev_order=25, n_pca=n_pca or None, decay=None if mode == "unweighted" else 40,
                       distance="cosine" if mode == "cosine" else "euclidean")
    if mode == "replicated":
        # a graph every rank holds in full (here: the oracle's, uploaded like a graph built elsewhere), rows sharded for the
        # recurrences only (meld_amd.distributed.shard_of_graph)
        from meld_amd.graph import DeviceGraph

        Go = mo.build_graph(X, knn=knn, algorithm="brute")
        W = Go.W.tocsr()
        W.sort_indices()
        full = DeviceGraph(torch.from_numpy(W.indptr.astype(np.int64)), torch.from_numpy(W.indices.astype(np.int32)),
                           torch.from_numpy(W.data.astype(np.float64)), torch.from_numpy(np.asarray(Go.dw, dtype=np.float64)), anisotropy=1.0)
        comm = mdist.Comm()
        op.graph = mdist.shard_of_graph(full, CpuOps(), comm)
        op.X = torch.from_numpy(X)
        dens = op.transform(labels)
        G = op.graph
        G.info.setdefault("exchange", "none")
        G.info.setdefault("exchange_overflow", 0)
    else:
        dens = mdist.fit_transform_sharded(op, torch.from_numpy(X), labels, ops=CpuOps(), comm=mdist.Comm())
    G = op.graph
    # the same estimate with two all-reduces per iteration (the unfolded phases): every collective again, on every rank
    os.environ["MELD_LANCZOS_FOLD"] = "0"
    from meld_amd.filter import lanczos_lmax

    theta2, info2 = lanczos_lmax(G, tol=G.lmax_info["tol"])
    os.environ["MELD_LANCZOS_FOLD"] = "1"
    np.savez(
        out_path + ".rank{}".format(dist.get_rank()), dens=dens.values, columns=np.asarray(dens.columns, dtype=str),
        lmax=G.lmax, lanczos_iters=G.lmax_info["iterations"],
        lanczos_all_reduces=G.lmax_info.get("all_reduces_per_iteration", 2), theta_unfolded=theta2, iters_unfolded=info2["iterations"], row_begin=G.row_begin, n_rows=G.n_rows,
        rowptr=G.rowptr.numpy(), col=G.col.numpy(), val=G.val.numpy(), dw=G.dw_dev.numpy(), nnz_global=G.info["nnz_global"], exchange=G.info["exchange"], exchange_overflow=G.info["exchange_overflow"],
    )
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]) if len(sys.argv) > 6 else 0)
