"""Does the order in which the search workgroups are dispatched matter?  Work per workgroup = tiles it stages (from the
pruning table and the final thresholds); list scheduling over the resident slots in index order vs longest-first.
python tools/sim_wg_schedule.py   (after tools/knn_ablate.py's setup; 1M cells)"""
import os, sys, math, heapq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
os.environ["MELD_KNN16_ABLATION"] = "99"   # (makes knn_ablate stop after its setup + one product run)
import numpy as np, torch
src = open(os.path.join(os.path.dirname(__file__), "knn_ablate.py")).read().split("abl = os.environ.get")[0]
exec(src)
run(mseed, "product")
thr = (cthr * float(sinfo[0]) ** 2).view(-1, 64).max(1).values            # final wmax per wave (scaled units)
margin = float(lib.meld_knn16_error_coef(1, d)) * float(nmax) * float(sinfo[0]) ** 2
tab = lb2.view(torch.float16).view(q_pad // 64, n_tiles)
work = torch.zeros(q_pad // 256, dtype=torch.int64, device="cuda")
for w0 in range(0, q_pad // 64, 4096):
    t = tab[w0:w0 + 4096].float()
    live = t <= (thr[w0:w0 + 4096, None] + margin)                         # [waves, tiles]
    uni = live.view(-1, 4, n_tiles).any(1).sum(1)                          # tiles the workgroup stages (union of its 4 waves)
    work[w0 // 4: w0 // 4 + uni.shape[0]] = uni
w = work.cpu().numpy().astype(np.float64)
print("workgroups %d, staged tiles: mean %.0f  min %.0f  max %.0f  (sum %.3g)" % (len(w), w.mean(), w.min(), w.max(), w.sum()))
def makespan(order, slots=768):
    h = [0.0] * slots
    heapq.heapify(h)
    for i in order:
        t = heapq.heappop(h)
        heapq.heappush(h, t + w[i])
    return max(h)
ideal = w.sum() / 768
print("ideal (perfect balance) %.0f tile-steps;  index order %.0f (+%.1f %%);  longest first %.0f (+%.1f %%)" % (
    ideal, makespan(range(len(w))), 100 * (makespan(range(len(w))) / ideal - 1), makespan(np.argsort(-w)), 100 * (makespan(np.argsort(-w)) / ideal - 1)))
