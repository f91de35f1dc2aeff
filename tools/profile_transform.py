"""Where does MELD.transform spend its time at 1M cells?  python tools/profile_transform.py [N]
Host wall times of the pieces of transform (device synchronised around each)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
import meld_amd
from meld_amd import filter as mf
from bench import synthetic_cells

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X, labels = synthetic_cells(N, 50, seed=0)
Xd = torch.from_numpy(X).cuda()


def t(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return 1e3 * best, out


for rep in range(2):
    op = meld_amd.MELD(knn=15, beta=60, chebyshev_order=30, verbose=0)
    ms_fit, _ = t(lambda: op.fit(Xd), 1)
    G = op.graph
    ms_lay, _ = t(lambda: G.ops.pt_layout(G) if G.ops else None, 1)
    ms_lmax, _ = t(lambda: G.estimate_lmax(), 1)
    ms_tr, out = t(lambda: op.transform(labels), 1)
    print("rep %d: fit %.2f ms | layout build %.2f | lmax %.2f (%d its) | transform (lmax known) %.2f" % (rep, ms_fit, ms_lay, ms_lmax, G.lmax_info["iterations"], ms_tr))
ms, fz = t(lambda: op._factorize_device(np.asarray(labels), G.val.device))
print("  label factorisation on the device: %.2f ms" % ms)
codes, samples, counts = fz
sig = mf.IndicatorSignal(codes, 2, 1.0 / np.asarray(counts, dtype=np.float64))
ms, s_dev = t(lambda: sig.to_device(G.val.device))
print("  indicator assembly: %.2f ms" % ms)
ms, sp = t(lambda: s_dev.index_select(0, G.perm))
print("  permute signal: %.2f ms" % ms)
c = mf.chebyshev_coefficients(mf.spectral_kernel("heat", 60, 0, 1, G.lmax), G.lmax, 30)
ms, r = t(lambda: mf.chebyshev_apply(G, sp, c, G.lmax))
print("  chebyshev_apply: %.2f ms" % ms)
def back():
    o = torch.empty_like(r); o[G.perm] = r
    st = torch.empty(o.shape, dtype=o.dtype, pin_memory=True); st.copy_(o, non_blocking=True); torch.cuda.current_stream().synchronize(); return st.numpy().copy()
ms, host = t(back)
print("  un-permute + D2H (fresh pinned buffer each time): %.2f ms" % ms)
import pandas as pd
t0 = time.perf_counter(); df = pd.DataFrame(host, index=None, columns=samples); print("  DataFrame: %.2f ms" % (1e3 * (time.perf_counter() - t0)))
ms, _ = t(lambda: mf.filter(sig, G, "heat", 60, chebyshev_order=30))
print("  filter() as a whole: %.2f ms" % ms)
