"""One-off exactness stress of the seeded pruning table + dispatch order: the graph with all search options on must equal
the graph of the plain search (no pruning, no seeds) bit for bit.  python tools/stress_pruning.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
from meld_amd.graph import HipOps
from meld_amd.reorder import locality_permutation

rng = np.random.default_rng(11)
cases = []
N = 300_000
cases.append(("gauss 10d in 30", rng.normal(size=(N, 10)) @ rng.normal(size=(10, 30)) + 0.01 * rng.normal(size=(N, 30))))
cases.append(("student-t 8d", rng.standard_t(2.5, size=(N, 8))))
cases.append(("clusters of very different scale 20d", np.concatenate([rng.normal(size=(N // 2, 20)) * 0.01, rng.normal(size=(N // 2, 20)) * 5.0 + 30.0])))
cases.append(("curve in 50d", np.stack([np.sin(np.linspace(0, 60, N) * (k + 1) / 7.0) for k in range(50)], 1) + 0.001 * rng.normal(size=(N, 50))))
cases.append(("uniform 4d", rng.random(size=(N, 4))))
for name, X in cases:
    Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).cuda()
    Xd = Xd.index_select(0, locality_permutation(Xd))
    outs = []
    for full in (False, True):
        ops = HipOps(prune=full)
        ops.seed = full
        keys, vals, bw, info = ops.directed_kernel_coo(Xd, 0, N, 15, 40, 1e-4, 64)
        outs.append(ops.assemble_rows(keys, vals, 0, N, N) + (bw,))
        if full:
            frac = info["wave_tiles_done"] / ((N + 63) // 64) ** 2
    same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    print("%-40s identical: %s   blocks computed %.3f  re-searched %d  swept %d" % (name, same, frac, info["n_researched_rows"], info["n_flagged_rows"]), flush=True)
    assert same
print("ok")
