"""Time the candidate-search stage alone on the benchmark cells (in locality order, as in fit):
python tools/knn_only.py N[,N...] [reps]      (env: MELD_KNN_* switches of HipOps, MELD_KNN16_ABLATION)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import torch

from meld_amd import graph as mg
from meld_amd.graph import HipOps
from meld_amd.reorder import locality_permutation
from oracle import meld_oracle as mo

reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for n in [int(a) for a in sys.argv[1].split(",")]:
    X, _ = mo.synthetic_cells(n, n_dims=int(os.environ.get("DIMS", "50")), seed=0)
    Xd = torch.from_numpy(X).cuda()
    perm = locality_permutation(Xd)
    if perm is not None:
        Xd = Xd.index_select(0, perm).contiguous()
    ops = HipOps()
    for r in range(reps):
        mg.record_events(True)
        q0, qc = int(os.environ.get("Q0", "0")), int(os.environ.get("QC", str(n)))  # (a row shard: queries [Q0, Q0 + QC) against all n cells)
        keys, vals, bw, info = ops.directed_kernel_coo(Xd, q0, qc, int(os.environ.get("KNN", "15")), 40, 1e-4, int(os.environ.get("KSEL", "64")))
        torch.cuda.synchronize()
        ev = mg.event_times_ms()
        ms = ev["knn_topk"][0]
        print({k: round(v[0], 2) for k, v in ev.items()})
        mg.record_events(False)
    kb = (int(os.environ.get("DIMS", "50")) + 3 + 15) // 16
    ideal = (n / 32.0) ** 2 * kb * 32 / 1024 / 2.4e9 * 1e3
    print("pairs listed-and-tested %s, blocks of 32 refs computed in full %s, two_phase %s" % (info.get("wave_tiles_done"), info.get("blocks_past_partial_test"), info.get("two_phase")))
    print("N=%d knn_topk %.2f ms  (hi.hi MFMA stream @2.4GHz %.2f ms, %.1f%%)  stage2 %.2f ms  researched %d  swept %d  env %s" % (
        n, ms, ideal, 100 * ideal / ms, ev.get("knn_topk_stage2", [0.0])[0], info["n_researched_rows"], info["n_flagged_rows"],
        {k: v for k, v in os.environ.items() if k.startswith("MELD_KNN")}))
