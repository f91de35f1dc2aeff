"""Helpers around the estimator (reference ``meld/utils.py``)."""
from __future__ import annotations

import numpy as np
import pandas as pd

__all__ = ["_check_pygsp_graph", "get_meld_cmap", "normalize_densities"]


def _foreign_weights(G):
    """The scipy-sparse weight matrix of a graph object built elsewhere (graphtools / pygsp graphs
    expose it as ``.W``), or None when ``G`` does not look like such a graph."""
    from scipy import sparse

    if isinstance(G, np.ndarray) or sparse.issparse(G):
        return None
    try:
        W = getattr(G, "W", None)
    except Exception:  # a property that fails is not a graph we can adopt
        return None
    if W is None or not sparse.issparse(W) or W.shape[0] != W.shape[1]:
        return None
    n = getattr(G, "N", None)
    if n is not None and int(n) != W.shape[0]:
        return None
    return W


def _check_pygsp_graph(G):
    """Type guard of reference ``meld/utils.py:11-20``.  The reference accepts graphtools graphs
    (converting non-PyGSP ones with ``to_pygsp()``); here the graph the filter runs on is the
    device-resident ``DeviceGraph``, and a graph built elsewhere -- anything exposing a square
    scipy-sparse ``.W`` (graphtools / pygsp graphs do; reference ``meld/benchmark.py:194-195``,
    ``test/test_utils.py:11-13``) -- is uploaded once (``DeviceGraph.from_scipy``), keeping an
    ``lmax`` the graph already carries.  Anything else raises the reference's ``TypeError``
    (message pinned by ``test/test_meld.py:22-28``)."""
    from .graph import DeviceGraph

    if isinstance(G, DeviceGraph):
        return G
    W = _foreign_weights(G)
    if W is not None:
        return DeviceGraph.from_foreign(G, W)
    raise TypeError(
        "Input graph should be of type graphtools.base.BaseGraph."
        " With graphtools, use the `use_pygsp=True` flag."
    )


def get_meld_cmap():
    """Colormap used in the publication for displaying EES (reference ``meld/utils.py:23-32``):
    blue - grey - red, linearly interpolated."""
    from matplotlib.colors import LinearSegmentedColormap

    stops = [
        [0.22107637, 0.53245276, 0.72819301, 1.0],
        [0.7, 0.7, 0.7, 1],
        [0.75013244, 0.3420382, 0.22753009, 1.0],
    ]
    return LinearSegmentedColormap.from_list("meld_cmap", stops)


def normalize_densities(sample_densities):
    """Row-normalise ``[N, p]`` sample densities so that each row sums (in absolute value) to 1
    -- the sample likelihoods (reference ``meld/utils.py:35-47``, sklearn ``normalize(norm='l1')``;
    all-zero rows are returned unchanged).  DataFrame index/columns are preserved.  Runs as a
    one-pass HIP kernel (``meld_normalize_rows_l1``)."""
    import torch

    from ._lib import check, get_lib, ptr

    is_df = isinstance(sample_densities, pd.DataFrame)
    if is_df:
        index, columns = sample_densities.index, sample_densities.columns
    arr = np.ascontiguousarray(np.asarray(sample_densities, dtype=np.float64))
    if arr.ndim != 2:
        raise ValueError("Expected 2D array, got {}D array instead".format(arr.ndim))
    lib = get_lib()
    if not torch.cuda.is_available():
        raise RuntimeError("meld_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
    x = torch.from_numpy(arr).to("cuda")
    out = torch.empty_like(x)
    check(
        lib.meld_normalize_rows_l1(ptr(x), ptr(out), arr.shape[0], arr.shape[1], torch.cuda.current_stream().cuda_stream),
        "meld_normalize_rows_l1",
    )
    res = out.cpu().numpy()
    if is_df:
        res = pd.DataFrame(res, index=index, columns=columns)
    return res
