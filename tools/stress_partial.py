"""Repeated builds of the same cells with and without the partial test (reference slices, mid-sized data): every build must give the
same bandwidths.  python tools/stress_partial.py [reps] [d,d,...]     (the regression this guards: tests/test_gpu_partial_search.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
os.environ.setdefault("MELD_KNN_ROTATE_MIN", "0")  # (the product takes the frame from 262144 cells on)
from meld_amd.graph import HipOps
from meld_amd.reorder import locality_permutation

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for d in [int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else "20,32,45,50,58,62").split(",")]:
    rng = np.random.default_rng(d)
    X = rng.normal(size=(33555, d)) * (10.0 ** rng.uniform(-3, 1, size=d))
    Xd = torch.from_numpy(X).cuda()
    Xd = Xd.index_select(0, locality_permutation(Xd)).contiguous()
    plain, framed = HipOps(), HipOps()
    plain.rotate = False

    def bw(ops, ee):
        if ee is None:
            os.environ.pop("MELD_KNN16_EE", None)
        else:
            os.environ["MELD_KNN16_EE"] = ee
        return ops.directed_kernel_coo(Xd, 0, 33555, 5, 40, 1e-2, 64)[2]

    ref = bw(plain, "0")
    bad = [0, 0, 0]
    for _ in range(reps):
        bad[0] += int(not torch.equal(bw(plain, "1"), ref))    # test forced on, cells as given
        bad[1] += int(not torch.equal(bw(framed, None), ref))  # the product: principal frame + test
        bad[2] += int(not torch.equal(bw(plain, "0"), ref))    # no test
    print("d=%d (%d K blocks): builds that differ, of %d each -- test forced on %d, principal frame + test %d, no test %d" % (d, (d + 3 + 15) // 16, reps, *bad), flush=True)
os.environ.pop("MELD_KNN16_EE", None)
