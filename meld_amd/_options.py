"""Every switch of the package in one place.

PUBLIC options -- the handful a deployment may set -- are read from the environment whenever they are asked for.  Everything else
is a DEVELOPMENT switch (A-B measurements, ablations, regression tests of paths the product no longer takes): it keeps its
default unless ``MELD_DEV=1`` is set, so that a stray variable in a production environment cannot change which kernels run.  The
library (``csrc/common.hpp``, ``meld_dev_getenv``) follows the same rule for the switches it reads itself.  ``tests/conftest.py``
and the scripts under ``tools/`` set ``MELD_DEV=1``; ``bench.py`` sets it only around its one untimed unpruned pass.

=====================  ==========  =======================================================================================
public option           default     meaning
=====================  ==========  =======================================================================================
MELD_HIP_LIB            (in-tree)   path of an alternative build of libmeld_hip.so
MELD_SPMM               auto        recurrence kernel: auto | tiled | csr
MELD_KNN_SEARCH         f16x3       candidate search: f16x3 (split-fp16 MFMA) | f32 (first generation)
MELD_KNN_ROTATE         1           search in the cells' principal frame where it pays (0: cells as given)
MELD_KNN_ROTATE_MIN     262144      ... from this many cells on (the frame costs ~0.9 ms whatever the size)
MELD_REORDER            1           cache-locality permutation of the cells before the build
MELD_PINNED_RESULT      1           densities are handed to the caller in the pinned buffer they left the device through
MELD_SHARDED_C_LOOPS    1           row-sharded recurrences enqueued from C on the library's own RCCL communicator
=====================  ==========  =======================================================================================
"""
import os

PUBLIC = frozenset({"MELD_HIP_LIB", "MELD_SPMM", "MELD_KNN_SEARCH", "MELD_KNN_ROTATE", "MELD_KNN_ROTATE_MIN", "MELD_REORDER", "MELD_PINNED_RESULT",
                    "MELD_SHARDED_C_LOOPS"})


def dev_mode():
    return os.environ.get("MELD_DEV", "0") == "1"


def opt(name, default=None):
    """Value of switch ``name``: the environment's for a public option, and for a development switch under ``MELD_DEV=1``;
    else ``default``."""
    if name in PUBLIC or dev_mode():
        return os.environ.get(name, default)
    return default


def is_set(name):
    return opt(name) is not None
