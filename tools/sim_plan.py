"""What a time-balanced row partition could give the recurrence kernel (1M benchmark graph, 256 row blocks): evaluates
the block-time model fitted in profiles/r03_recurrence_step_timeline.txt,
    t = 2.2 + 0.361 E/1000 + 1.077 D/1000 + 2.840 R/1000  us   (E entries, D distinct out-of-block columns, R rows),
for (a) the product's plan (equal entries, at most RMAX rows), (b) a plan by per-row weights that need no knowledge of the
blocks -- a deg_i + b far_i + c, far_i = neighbours further than `win` rows away --, (c) plans refined by the exact D of the
previous plan (what a two-pass build could do), each with RMAX = 4080 and larger.
python tools/sim_plan.py graph.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import torch

d = torch.load(sys.argv[1])
rowptr, col = d["rowptr"].cuda(), d["col"].cuda().to(torch.int64)
N = int(d["N"]); nnz = col.numel(); NB = 256
deg = (rowptr[1:] - rowptr[:-1]).double()
rows = torch.repeat_interleave(torch.arange(N, device="cuda"), rowptr[1:] - rowptr[:-1])
A, B, C, T0 = 0.361e-3, 1.077e-3, 2.840e-3, 2.2
if os.environ.get("MODEL"):  # "a,b,c,t0" of a newer fit (us per 1000 entries / distinct columns / rows, constant)
    a_, b_, c_, t_ = (float(v) for v in os.environ["MODEL"].split(","))
    A, B, C, T0 = a_ * 1e-3, b_ * 1e-3, c_ * 1e-3, t_


def cuts_from_weights(w, rmax):
    """contiguous blocks with equal weight, at most rmax rows each (the product's cap rule)"""
    pre = torch.cumsum(w, 0)
    tot = float(pre[-1])
    targets = torch.arange(1, NB + 1, device="cuda", dtype=torch.float64) * (tot / NB)
    cut = torch.searchsorted(pre, targets).clamp_(max=N).tolist()
    cut[-1] = N
    out, prev = [0], 0
    for b in range(NB):
        c = max(cut[b], N - (NB - 1 - b) * rmax)
        c = min(c, prev + rmax)
        c = min(max(c, prev), N)
        out.append(c); prev = c
    return torch.tensor(out, device="cuda")


def evaluate(cuts, label):
    blk = torch.bucketize(rows, cuts[1:], right=True)       # block of every entry's row
    cblk = torch.bucketize(col, cuts[1:], right=True)
    E = torch.bincount(blk, minlength=NB).double()
    out = blk != cblk
    key = torch.unique(blk[out] * N + col[out])
    D = torch.bincount(key // N, minlength=NB).double()
    R = (cuts[1:] - cuts[:-1]).double()
    t = T0 + A * E + B * D + C * R
    print("%-58s max %.1f  mean %.1f  p90 %.1f  (rows max %d, D max %d)" % (label, float(t.max()), float(t.mean()), float(t.quantile(0.9)), int(R.max()), int(D.max())))
    return t, E, D, R


if os.environ.get("GRID"):  # planner weights (per entry, per far entry, per row) scored under the model, RMAX = 4080
    far = {}
    for win in (1000, 2000, 3000):
        f = torch.zeros(N, device="cuda", dtype=torch.float64)
        f.index_add_(0, rows, ((col - rows).abs() > win).double())
        far[win] = f
    evaluate(cuts_from_weights(0.361 * deg + 0.754 * far[2000] + 2.84, 4080), "product weights 0.361 deg + 0.754 far(>2000) + 2.84")
    res = []
    for win in far:
        for wf in (0.4, 0.6, 0.8, 1.0, 1.3):
            for wr in (2.0, 3.5, 5.0, 7.0):
                w = A * 1e3 * deg + wf * far[win] + wr
                cuts = cuts_from_weights(w, 4080)
                blk = torch.bucketize(rows, cuts[1:], right=True); cblk = torch.bucketize(col, cuts[1:], right=True)
                E = torch.bincount(blk, minlength=NB).double(); out = blk != cblk
                D = torch.bincount(torch.unique(blk[out] * N + col[out]) // N, minlength=NB).double()
                R = (cuts[1:] - cuts[:-1]).double()
                t = T0 + A * E + B * D + C * R
                res.append((float(t.max()), float(t.quantile(0.9)), win, wf, wr))
    for r in sorted(res)[:12]:
        print("max %.1f p90 %.1f   win %d  w_far %.2f  w_row %.1f  (w_entry %.3f)" % (r + (A * 1e3,)))
    sys.exit(0)
for rmax in (4080, 4608, 5120):
    c0 = cuts_from_weights(deg, rmax)
    t, E, D, R = evaluate(c0, "RMAX %d: equal entries (product)" % rmax)
    for win in (2000, 4000):
        far = torch.zeros(N, device="cuda", dtype=torch.float64)
        far.index_add_(0, rows, ((col - rows).abs() > win).double())
        for bscale in (0.5, 0.7, 1.0):
            w = A * deg + B * bscale * far + C
            evaluate(cuts_from_weights(w, rmax), "RMAX %d: weights a deg + %.1f b far(>%d) + c" % (rmax, bscale, win))
    # refinement by the exact D of the previous plan: time density per row = t_b / R_b
    cuts = c0
    for it in range(3):
        dens = (t - T0) / R.clamp(min=1)
        blk_of_row = torch.bucketize(torch.arange(N, device="cuda"), cuts[1:], right=True)
        cuts = cuts_from_weights(dens[blk_of_row], rmax)
        t, E, D, R = evaluate(cuts, "RMAX %d: refined by exact D, round %d" % (rmax, it + 1))
