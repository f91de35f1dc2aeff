"""meld_amd -- MI355X-native implementation of the MELD density-estimation hot path.

Drop-in for the ``meld`` package on that path: ``import meld_amd as meld`` then
``meld.MELD().fit_transform(X, sample_labels)`` (reference ``meld/__init__.py:3-8``,
``README.md:46-61``).
"""
from .version import __version__
from .meld import MELD
from .utils import get_meld_cmap, normalize_densities
from . import utils
from . import filter  # noqa: A004
from .graph import DeviceGraph, build_knn_graph

__all__ = [
    "MELD",
    "DeviceGraph",
    "build_knn_graph",
    "get_meld_cmap",
    "normalize_densities",
    "VertexFrequencyCluster",
    "Benchmarker",
    "utils",
    "filter",
    "__version__",
]


def __getattr__(name):
    # exported by the reference package but outside the hot path
    if name == "VertexFrequencyCluster":  # lazy: pulls in the dense linear-algebra path only when used
        from .cluster import VertexFrequencyCluster

        return VertexFrequencyCluster
    if name == "Benchmarker":  # host-side helper of the reference package (meld/benchmark.py), lazy like the above
        from .benchmark import Benchmarker

        return Benchmarker
    raise AttributeError("module 'meld_amd' has no attribute {!r}".format(name))
