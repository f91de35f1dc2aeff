import os
import sys

import pytest

# the tests force paths the product no longer takes by default (MELD_KNN_ROTATE_MIN, MELD_KNN16_EE, MELD_ASSEMBLE ...): development
# switches, read only under MELD_DEV=1 (meld_amd/_options.py); worker processes inherit it
os.environ.setdefault("MELD_DEV", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
