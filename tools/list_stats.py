"""Statistics of the step lists of the first pass (meld_knn16_step_lists): how many waves of a block need a listed tile,
how long a wave's runs of needed / not needed steps are.   python tools/list_stats.py [N]"""
import os, sys
os.environ["MELD_KNN16_ABLATION"] = "99"  # (knn_ablate: set-up only)
sys.argv = [sys.argv[0]] + sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import knn_ablate as ka  # noqa: E402  (runs the set-up: operands, seeds, table, lists)

sl, sc = ka.LISTS
n_tiles = ka.n_tiles
nb = sc.shape[0]
sl = sl.view(nb, n_tiles)
cnt = sc.to(torch.int64)
print("blocks %d, steps per block: mean %.0f  min %d  max %d  (tiles %d)" % (nb, float(cnt.float().mean()), int(cnt.min()), int(cnt.max()), n_tiles))
idx = torch.arange(n_tiles, device="cuda").unsqueeze(0)
valid = idx < cnt.unsqueeze(1)
mask = (sl >> 24) & 0xF
pc = ((mask & 1) + ((mask >> 1) & 1) + ((mask >> 2) & 1) + ((mask >> 3) & 1))
tot = int(valid.sum())
for k in range(5):
    print("steps needed by %d waves: %.1f %%" % (k, 100.0 * int(((pc == k) & valid).sum()) / tot))
print("live wave-steps: %.1f %%" % (100.0 * int((pc * valid).sum()) / (4 * tot)))
# run lengths of wave 0's bit along the lists of 200 sampled blocks
import numpy as np
runs_on, runs_off = [], []
for b in np.linspace(0, nb - 1, 200).astype(int):
    m = ((mask[b, : int(cnt[b])] >> 0) & 1).cpu().numpy()
    if len(m) == 0: continue
    ch = np.flatnonzero(np.diff(m)) + 1
    seg = np.diff(np.concatenate([[0], ch, [len(m)]]))
    vals = m[np.concatenate([[0], ch])]
    runs_on += list(seg[vals == 1]); runs_off += list(seg[vals == 0])
print("wave 0: runs of needed steps: mean %.1f median %d;  runs of not-needed steps: mean %.1f median %d  p90 %d" % (
    np.mean(runs_on), np.median(runs_on), np.mean(runs_off), np.median(runs_off), np.percentile(runs_off, 90)))
# per-block imbalance: steps of the block vs the live steps of its busiest / its average wave
live = torch.stack([(((mask >> w) & 1) * valid).sum(1) for w in range(4)], 1).float()
print("per block: steps / busiest wave's live steps = %.3f;  steps / mean wave's live steps = %.3f" % (
    float((cnt.float() / live.max(1).values.clamp(min=1)).mean()), float((cnt.float() / live.mean(1).clamp(min=1)).mean())))
print("sum over blocks: steps %.3e, busiest-wave live steps %.3e, mean-wave live steps %.3e" % (float(cnt.sum()), float(live.max(1).values.sum()), float(live.mean(1).sum())))
