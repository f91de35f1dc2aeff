"""Worker of tests/test_distributed_gloo.py::test_sharded_filterbank_vfc: the filter-bank VertexFrequencyCluster on a
row-sharded graph (gloo, CPU tensors, NumPy stand-in for the kernels)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path, n, d, knn):
    dist.init_process_group("gloo")
    import meld_amd
    from meld_amd import distributed as mdist
    from meld_amd.graph import DeviceGraph
    from oracle import meld_oracle as mo
    from tests.cpu_ops import CpuOps

    X, labels = mo.synthetic_cells(n, n_dims=d, seed=7)
    op = meld_amd.MELD(knn=knn, beta=40, chebyshev_order=25)
    dens = mdist.fit_transform_sharded(op, torch.from_numpy(X), labels, ops=CpuOps(), comm=mdist.Comm())
    G = op.graph
    kw = dict(method="filterbank", n_probes=int(os.environ.get("MELD_TEST_PROBES", "24")), n_bands=6, window_sizes=np.array([1, 2, 4, 8]), chebyshev_order=48, random_state=3, n_clusters=3)
    vfc = meld_amd.VertexFrequencyCluster(**kw)
    vfc.fit(G)
    out = dict(spec=vfc._fb_spectrogram.numpy(), norm2=vfc._fb["window_norm2"].numpy(), ritz=vfc._fb["ritz"].numpy(), lmax=G.lmax)
    if dist.get_world_size() == 1:
        # the same algorithm on the same rows without a process group (the single-GPU code path, on CPU tensors here)
        G1 = DeviceGraph(G.rowptr, G.col, G.val, G.dw_dev, ksum=G.ksum[: G.N], anisotropy=G.anisotropy)
        G1.ops = CpuOps()
        G1.lmax = G.lmax
        v1 = meld_amd.VertexFrequencyCluster(**kw)
        v1.fit(G1)
        out["spec_unsharded"] = v1._fb_spectrogram.numpy()
    np.savez(out_path + ".rank{}".format(dist.get_rank()), **out)
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
