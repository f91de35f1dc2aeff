"""Where does a step of the recurrence kernel spend its time?  Per-wave wall-clock stamps of one launch
(meld_pt_debug_stamps): python tools/spmm_stamps.py graph.pt [p]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
from meld_amd.graph import DeviceGraph, HipOps
from meld_amd._lib import get_lib, ptr

d = torch.load(sys.argv[1])
pp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
G = DeviceGraph(d["rowptr"].cuda(), d["col"].cuda(), d["val"].cuda(), d["dw"].cuda())
G.ops = HipOps(spmm="tiled")
pt = G.ops.pt_layout(G)
nb = pt["nb"]
x = torch.rand(G.N, pp, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
r = torch.zeros_like(x)
lib = get_lib()
mask = int(os.environ.get("PT_MASK", "0"))
if mask:
    lib.meld_pt_debug_ablate(mask)
for _ in range(5):
    G.ops.cheby_step(G, pp, x, 0, y, y, r, 1e-3, 0.5, 0.5, 0.0)
buf = torch.zeros(nb * 16 * 8, dtype=torch.int64, device="cuda")
lib.meld_pt_debug_stamps(ptr(buf))
G.ops.cheby_step(G, pp, x, 0, y, y, r, 1e-3, 0.5, 0.5, 0.0)
torch.cuda.synchronize()
lib.meld_pt_debug_stamps(None)
t = buf.cpu().numpy().reshape(nb, 16, 8).astype(np.float64) / 100.0  # us
t0 = t[:, :, 0].min()
t -= t0
names = ["start", "pre-barrier", "post-barrier", "IN done", "stream done", "pre-close", "post-close", "written"]
print("nb %d; kernel span %.1f us (first start -> last written)" % (nb, t[:, :, 7].max()))
cons, load = t[:, :12, :], t[:, 12:, :]
for i, nm in enumerate(names):
    c = cons[:, :, i]
    print("  %-13s consumers: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % ((nm,) + tuple(np.percentile(c, [0, 10, 50, 90, 100]))))
l = load[:, :, 5]
print("  loaders done (pre-close): median %.1f max %.1f" % (np.median(l), l.max()))
blk_end = t[:, :, 7].max(axis=(1))
blk_start = t[:, :, 0].min(axis=1)
print("  block start: min %.1f median %.1f max %.1f;  block end: min %.1f median %.1f max %.1f" % (blk_start.min(), np.median(blk_start), blk_start.max(), blk_end.min(), np.median(blk_end), blk_end.max()))
sd = cons[:, :, 4]
print("  stream-done spread within a block (max - min over its 12 waves): median %.1f max %.1f" % (np.median(sd.max(1) - sd.min(1)), (sd.max(1) - sd.min(1)).max()))
ind = cons[:, :, 3]
print("  IN-done spread within a block: median %.1f max %.1f; IN duration (post-barrier -> IN done) median %.1f" % (np.median(ind.max(1) - ind.min(1)), (ind.max(1) - ind.min(1)).max(), np.median(ind - cons[:, :, 2])))
print("  OUT duration (IN done of the slowest wave -> stream done) median %.1f" % np.median(sd.max(1) - ind.max(1)))
nt = pt["tensors"]["blk_ntile"].cpu().numpy(); br = pt["tensors"]["blk_row"].cpu().numpy()
print("  tiles/block mean %.1f max %d; rows/block %d..%d" % (nt.mean(), nt.max(), np.diff(br).min(), np.diff(br).max()))
# what makes a block slow?  least squares of its duration on (entries, distinct OUT columns, rows)
rp = G.rowptr.cpu().numpy()
ent = np.diff(rp[br]).astype(np.float64)
nd = pt["tensors"]["blk_ndist"].cpu().numpy().astype(np.float64)
nr = np.diff(br).astype(np.float64)
dur = blk_end - blk_start
A = np.stack([np.ones(nb), ent, nd, nr], 1)
coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
pred = A @ coef
print("  duration ~ %.1f + %.3f us per 1000 entries + %.3f us per 1000 distinct OUT columns + %.3f us per 1000 rows; residual rms %.2f us (duration rms spread %.2f)"
      % (coef[0], 1e3 * coef[1], 1e3 * coef[2], 1e3 * coef[3], np.sqrt(np.mean((pred - dur) ** 2)), dur.std()))
print("  entries/block %d..%d (mean %.0f); distinct OUT cols/block %d..%d (mean %.0f)" % (ent.min(), ent.max(), ent.mean(), nd.min(), nd.max(), nd.mean()))
for q in (0, 10, 50, 90, 100):
    i = np.argsort(dur)[min(nb - 1, int(q / 100 * nb))]
    print("   p%-3d block %3d: %.1f us  entries %d  ndist %d  rows %d  tiles %d" % (q, i, dur[i], ent[i], nd[i], nr[i], nt[i]))
