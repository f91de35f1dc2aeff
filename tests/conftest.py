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


def run_ranks(worker, args, world, timeout=900, env=None, attempts=2):
    """``torch.distributed.run`` of tests/<worker> on ``world`` local ranks (rendezvous on 127.0.0.1, a free port).  A launch that
    fails is repeated ONCE on a fresh port and the first failure is shown as a warning: the port is picked by binding and closing
    a socket, which any other process may take in between, and a rendezvous that times out says nothing about the code under
    test (seen once in ~300 launches on the GPU boxes; the comparison itself is deterministic).  A second failure fails the test
    with both outputs."""
    import socket
    import subprocess
    import warnings

    failures = []
    for attempt in range(attempts):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "tests", worker)] + [str(a) for a in args]
        res = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="2", **(env or {})), capture_output=True, text=True, timeout=timeout)
        if res.returncode == 0:
            return res
        failures.append("--- attempt %d, return code %d ---\n%s\n%s" % (attempt + 1, res.returncode, res.stdout[-3000:], res.stderr[-3000:]))
        if attempt + 1 < attempts:
            warnings.warn("launch of %s on %d ranks failed, repeating it once:\n%s" % (worker, world, failures[-1][-1500:]))
    raise AssertionError("\n".join(failures))
