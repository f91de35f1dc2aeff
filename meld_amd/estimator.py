"""Validated-attribute estimator base: the slice of ``graphtools.estimator.GraphEstimator`` that
``meld.MELD`` inherits (reference ``meld/meld.py:9,13,42-92,117-118``; [UPSTREAM graphtools 1.5.x
``estimator.attribute`` / ``GraphEstimator``]), re-stated for a device-resident graph.

Behaviour kept:
  * ``attribute(name, default, on_set=validators)`` properties that validate on assignment and
    raise ``ValueError`` with graphtools' messages (pinned by reference ``test/test_meld.py:100-105``);
  * graph parameters ``knn=5, decay=40, n_pca=100, thresh=1e-4, distance="euclidean", n_jobs,
    random_state, verbose`` with their defaults;
  * ``fit(X)`` adopts a prebuilt graph or builds one; refitting on different data rebuilds;
  * ``set_params`` of a graph parameter drops the graph (reference ``test/test_meld.py:90-93``).
"""
from __future__ import annotations

import numbers
from functools import partial

import numpy as np

__all__ = ["attribute", "GraphEstimator", "check_positive", "check_int", "check_in", "check_if_not"]


def check_positive(**params):
    for p in params:
        if not isinstance(params[p], numbers.Number) or params[p] <= 0:
            raise ValueError("Expected {} > 0, got {}".format(p, params[p]))


def check_int(**params):
    for p in params:
        if not isinstance(params[p], numbers.Integral):
            raise ValueError("Expected {} integer, got {}".format(p, params[p]))


def check_in(choices, **params):
    for p in params:
        if params[p] not in choices:
            raise ValueError("{} value {} not recognized. Choose from {}".format(p, params[p], choices))


def check_if_not(x, *checks, **params):
    for p in params:
        if params[p] is not x and params[p] != x:
            for chk in checks:
                chk(**{p: params[p]})


def attribute(attr, default=None, doc=None, on_set=None):
    """Property stored as ``_<attr>`` whose setter runs the ``on_set`` validators."""
    validators = [] if on_set is None else ([on_set] if callable(on_set) else list(on_set))

    def fget(self):
        return getattr(self, "_" + attr, default)

    def fset(self, value):
        for fn in validators:
            fn(**{attr: value})
        setattr(self, "_" + attr, value)

    return property(fget=fget, fset=fset, doc=doc)


_GRAPH_PARAMS = ("knn", "decay", "n_pca", "thresh", "distance", "anisotropy", "n_landmark")
_PASSIVE_PARAMS = ("n_jobs", "random_state", "verbose")


class GraphEstimator(object):
    """Estimator that owns a graph built from data."""

    X = attribute("X", doc="Stored input data")
    n_pca = attribute("n_pca", default=100, on_set=partial(check_if_not, None, check_positive, check_int))
    random_state = attribute("random_state")
    knn = attribute("knn", default=5, on_set=[check_positive, check_int])
    decay = attribute("decay", default=40, on_set=partial(check_if_not, None, check_positive))
    distance = attribute("distance", default="euclidean", on_set=partial(check_in, ["euclidean", "l2", "sqeuclidean", "cosine", "correlation", "manhattan", "cityblock", "l1", "chebyshev",
                                                                           "precomputed", "precomputed_distance", "precomputed_affinity"]))
    n_jobs = attribute("n_jobs", default=1, on_set=check_int)
    verbose = attribute("verbose", default=0)
    thresh = attribute("thresh", default=1e-4, on_set=partial(check_if_not, 0, check_positive))
    n_landmark = attribute("n_landmark", on_set=partial(check_if_not, None, check_positive, check_int))

    def __init__(
        self,
        knn=5,
        decay=40,
        n_pca=100,
        n_landmark=None,
        random_state=None,
        verbose=0,
        n_jobs=1,
        distance="euclidean",
        thresh=1e-4,
        anisotropy=0,
        **kwargs
    ):
        if verbose is True:
            verbose = 1
        elif verbose is False:
            verbose = 0
        self.n_pca = n_pca
        self.n_landmark = n_landmark
        self.random_state = random_state
        self.knn = knn
        self.decay = decay
        self.distance = distance
        self.n_jobs = n_jobs
        self.verbose = verbose
        self.thresh = thresh
        self.anisotropy = anisotropy
        self.kwargs = kwargs
        self._graph = None

    # graph property: dropping the graph resets everything derived from it
    @property
    def graph(self):
        return getattr(self, "_graph", None)

    @graph.setter
    def graph(self, G):
        self._graph = G
        if G is None:
            self._reset_graph()

    def _reset_graph(self):  # overridden by subclasses
        pass

    def set_params(self, **params):
        for p, v in params.items():
            if p in _GRAPH_PARAMS:
                if getattr(self, p) != v:
                    setattr(self, p, v)
                    self.graph = None
            elif p in _PASSIVE_PARAMS:
                setattr(self, p, v)
            elif p in self.kwargs or p in ("ksel",):
                if self.kwargs.get(p) != v:
                    self.kwargs[p] = v
                    self.graph = None
            else:
                raise ValueError("Invalid parameter {} for estimator {}".format(p, type(self).__name__))
        return self

    def _log(self, msg):
        if self.verbose:
            print(msg, flush=True)

    def fit(self, X, **kwargs):
        """Build (or adopt) the graph.  ``X``: array-like [n_samples, n_features] (ndarray,
        DataFrame, anything with ``.X`` like AnnData) or an already built ``DeviceGraph``."""
        from . import graph as _graph

        from .utils import _foreign_weights

        if isinstance(X, _graph.DeviceGraph) or _foreign_weights(X) is not None:
            # a prebuilt graph (reference meld/benchmark.py:194-195): ours, or one built elsewhere
            # (graphtools / pygsp: uploaded once by utils._check_pygsp_graph)
            from .utils import _check_pygsp_graph

            self._log("Using precomputed graph and diffusion operator...")
            self.X = None
            self.graph = _check_pygsp_graph(X)
            self._graph_precomputed = True
            return self
        if getattr(self, "_graph_precomputed", False):
            # raw data after an adopted graph: that graph says nothing about these cells
            self.graph = None
            self._graph_precomputed = False
        try:
            import torch
        except ImportError:  # pragma: no cover
            torch = None
        if torch is not None and isinstance(X, torch.Tensor):
            # device-resident input: no host copy, no host-side equality check
            if X.dim() != 2:
                raise ValueError("Expected a 2D data matrix, got shape {}".format(tuple(X.shape)))
            if self.X is not X:
                self.graph = None
            self.X = X
            if self.graph is None:
                self._log("Building graph on {} samples and {} features.".format(X.shape[0], X.shape[1]))
                self.graph = self._build_timed(X, **kwargs)
            return self
        if hasattr(X, "X") and not isinstance(X, np.ndarray):  # AnnData-like
            X = X.X
        if hasattr(X, "sparse") and hasattr(X.sparse, "to_dense"):
            X = X.sparse.to_dense()
        data = np.asarray(getattr(X, "values", X))
        if hasattr(data, "toarray"):
            data = data.toarray()
        if data.ndim != 2:
            raise ValueError("Expected a 2D data matrix, got shape {}".format(data.shape))
        # (float32 input -- PCA scores usually are -- crosses PCIe as float32, half the bytes, and is widened on the device:
        # exact; converting 50 M entries on the host first costs more than the whole graph build)
        data = np.ascontiguousarray(data, dtype=np.float32 if data.dtype == np.float32 else np.float64)
        # (NaN / infinity are rejected by _build_graph once the data is on the device: a host-side
        # np.isfinite pass over 1M x 50 doubles costs ~40 ms, the device one 0.2 ms)
        if self.X is not None and (
            not isinstance(self.X, np.ndarray) or self.X.shape != data.shape or not np.array_equal(self.X, data)
        ):
            self.graph = None  # new data: rebuild
        self.X = data
        if self.graph is None:
            self._log("Building graph on {} samples and {} features.".format(data.shape[0], data.shape[1]))
            self.graph = self._build_timed(data, **kwargs)
        return self

    # stage names of the builder's timers -> the task names graphtools logs through tasklogger
    _STAGE_TASKS = (
        ("KNN search", ("reorder", "prepare", "bounds", "seed", "knn_topk", "refine", "knn_stage2", "radius_exact")),
        ("affinities", ("coo_emit", "symmetrize", "anisotropy_degree")),
    )

    def _build_timed(self, data, **kwargs):
        """``_build_graph`` with, when ``verbose``, the per-stage lines graphtools prints through tasklogger
        ("Calculating graph and diffusion operator... / Calculated KNN search in 0.06 seconds." -- e.g.
        ``notebooks/MELD_Quickstart.ipynb:190-198`` of the reference); the stage times come from the builder's own
        timers (a device synchronisation per stage, so only taken when asked for)."""
        import time

        if not self.verbose:
            return self._build_graph(data, **kwargs)
        self._log("Calculating graph and diffusion operator...")
        t0 = time.perf_counter()
        kw = dict(kwargs)
        kw.setdefault("profile", True)
        G = self._build_graph(data, **kw)
        stages = dict(getattr(G, "info", {}).get("stage_seconds", {}) or {})
        for task, names in self._STAGE_TASKS:
            t = sum(stages.get(k, 0.0) for k in names)
            if any(k in stages for k in names):
                self._log("  Calculating {}...".format(task))
                self._log("  Calculated {} in {:.2f} seconds.".format(task, t))
        self._log("Calculated graph and diffusion operator in {:.2f} seconds.".format(time.perf_counter() - t0))
        return G

    def _build_graph(self, data, **kwargs):
        raise NotImplementedError
