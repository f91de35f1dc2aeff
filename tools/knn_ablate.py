"""Where the pruned search spends its time: run it once, feed the final per-row thresholds back as seeds and time the
timing-only ablations of the kernel with realistic pruning (MELD_KNN16_ABLATION: 1 = no selection, 3 = MFMAs only,
9 = no tile loads).   python tools/knn_ablate.py [N]"""
import os, sys, math, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
from meld_amd._lib import get_lib, ptr, check
from meld_amd.reorder import locality_permutation
from bench import synthetic_cells

lib = get_lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X, _ = synthetic_cells(n, 50, seed=0)
Xd = torch.from_numpy(X).cuda()
Xd = Xd.index_select(0, locality_permutation(Xd)).contiguous()
N, d = Xd.shape
st = torch.cuda.current_stream().cuda_stream
TS, BQ = lib.meld_knn16_tile_refs(), lib.meld_knn16_block_queries()
ksel, knn = 64, 15; cap = lib.meld_knn16_row_capacity(ksel)
sums = torch.empty(d, dtype=torch.float64, device="cuda"); check(lib.meld_col_sums_f64(ptr(Xd), N, d, ptr(sums), st)); mean = sums / N
n_tiles = (N + TS - 1) // TS; q_pad = ((N + BQ - 1) // BQ) * BQ
Rt = torch.empty(n_tiles * lib.meld_knn16_tile_bytes(d), dtype=torch.uint8, device="cuda")
Q = torch.empty(q_pad * lib.meld_knn16_query_bytes(d), dtype=torch.uint8, device="cuda"); Qn = torch.empty(q_pad, dtype=torch.float32, device="cuda")
norm2 = torch.empty(N, dtype=torch.float32, device="cuda"); nmax = torch.zeros(1, dtype=torch.float32, device="cuda"); sinfo = torch.empty(4, dtype=torch.float32, device="cuda")
check(lib.meld_knn16_prepare(ptr(Xd), N, d, ptr(mean), 0, N, ptr(Rt), ptr(Q), ptr(Qn), ptr(norm2), ptr(nmax), ptr(sinfo), st))
tmpb = torch.empty(lib.meld_knn16_bounds_temp_bytes(N, d, N), dtype=torch.uint8, device="cuda")
lb2 = torch.empty(lib.meld_knn16_bounds_bytes(N, N), dtype=torch.uint8, device="cuda")
mseed = torch.empty(q_pad, dtype=torch.float32, device="cuda")   # the product's seeds (neighbourhood of the own block, on the matrix pipe)
check(lib.meld_knn16_seed_thresholds_mfma(ptr(Q), ptr(Qn), ptr(Rt), ptr(sinfo), ptr(nmax), N, d, 0, N, 15, (-math.log(1e-4)) ** (1 / 40), 1, 0, ptr(mseed), st))
seeded_table = os.environ.get("MELD_KNN_SEEDED_BOUNDS", "1") != "0"
check(lib.meld_knn16_bounds(ptr(Xd), N, d, ptr(mean), ptr(sinfo), ptr(nmax), ptr(Rt), 0, N, ptr(mseed) if seeded_table else None, ptr(Qn) if seeded_table else None, 1, ptr(tmpb), ptr(lb2), st))
ci = torch.empty(q_pad * cap, dtype=torch.int32, device="cuda"); cd = torch.empty(q_pad * cap, dtype=torch.float32, device="cuda"); cc = torch.empty(q_pad, dtype=torch.int32, device="cuda")
cthr = torch.full((q_pad,), float("inf"), dtype=torch.float32, device="cuda"); done = torch.zeros(1, dtype=torch.int64, device="cuda")
rf = (-math.log(1e-4)) ** (1 / 40)

ORDER = None
if os.environ.get("MELD_KNN_BLOCK_ORDER", "1") != "0":
    work = torch.empty(q_pad // BQ, dtype=torch.int32, device="cuda")
    check(lib.meld_knn16_block_work(ptr(lb2), ptr(mseed), N, d, N, 1, ptr(nmax), ptr(sinfo), ptr(work), st))
    ORDER = torch.argsort(work, descending=True, stable=True).to(torch.int32)

LISTS = None
if os.environ.get("MELD_KNN_STEP_LISTS", "1") != "0":  # the list-driven first pass (meld_knn16_step_lists / _topk_listed), as in the product
    sl = torch.empty((q_pad // BQ) * n_tiles, dtype=torch.int32, device="cuda"); sc = torch.empty(q_pad // BQ, dtype=torch.int32, device="cuda")
    check(lib.meld_knn16_step_lists(ptr(lb2), ptr(mseed), N, d, N, 1, ptr(nmax), ptr(sinfo), 0, ptr(sl), n_tiles, ptr(sc), st))
    LISTS = (sl, sc)
    if ORDER is not None:
        ORDER = torch.argsort(sc, descending=True, stable=True).to(torch.int32)

def run(seed, label):
    done.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if LISTS is not None:
        check(lib.meld_knn16_topk_listed(ptr(Q), ptr(Qn), ptr(Rt), ptr(sinfo), N, d, N, ksel, ptr(LISTS[0]), ptr(LISTS[1]), n_tiles, ptr(nmax), 0, ptr(seed),
                                         knn, rf, ptr(ci), ptr(cd), ptr(cc), ptr(cthr), ptr(done), ptr(ORDER) if ORDER is not None else None, 1, st))
    else:
      check(lib.meld_knn16_topk(ptr(Q), ptr(Qn), ptr(Rt), ptr(sinfo), N, d, N, ksel, 1, 1, ptr(lb2), ptr(nmax), 0, ptr(seed) if seed is not None else None,
                              knn, rf, ptr(ci), ptr(cd), ptr(cc), ptr(cthr), ptr(done), ptr(ORDER) if ORDER is not None else None, st))
    e1.record(); torch.cuda.synchronize()
    print("%-44s %.2f ms   blocks computed %.3f" % (label, e0.elapsed_time(e1), float(done) / ((q_pad // 64) * n_tiles)))

abl = os.environ.get("MELD_KNN16_ABLATION")
if abl is None:
    run(mseed, "product (seeds of the own neighbourhood)"); run(mseed, "product (seeds of the own neighbourhood)")
    seed = torch.minimum(cthr * float(sinfo[0]) ** 2 * 1.0001, mseed).contiguous()
    torch.save(seed.cpu(), "/tmp/knn_seed.pt")
    run(seed, "product, thresholds seeded with the final ones")
    print("seed / final threshold: median %.2f, p90 %.2f" % (float((mseed[:N] / seed[:N]).median()), float(torch.quantile((mseed[:N] / seed[:N])[:200000], 0.9))))
    if LISTS is not None:
        # what perfect seeds would be worth: table (per-query test) and lists rebuilt from the FINAL thresholds
        check(lib.meld_knn16_bounds(ptr(Xd), N, d, ptr(mean), ptr(sinfo), ptr(nmax), ptr(Rt), 0, N, ptr(seed), ptr(Qn), 1, ptr(tmpb), ptr(lb2), st))
        check(lib.meld_knn16_step_lists(ptr(lb2), ptr(seed), N, d, N, 1, ptr(nmax), ptr(sinfo), 0, ptr(LISTS[0]), n_tiles, ptr(LISTS[1]), st))
        if ORDER is not None:
            ORDER = torch.argsort(LISTS[1], descending=True, stable=True).to(torch.int32)
        run(seed, "table + lists + thresholds from the final ones"); run(seed, "table + lists + thresholds from the final ones")
        wm = seed[: (N // 64) * 64].view(-1, 64)
        fm = (cthr[: (N // 64) * 64] * float(sinfo[0]) ** 2).view(-1, 64)
        print("per wave: max seed / max final threshold: median %.2f p90 %.2f;  (product seeds) max seed / max final: median %.2f p90 %.2f" % (
            float((wm.max(1).values / fm.max(1).values).median()), float(torch.quantile(wm.max(1).values / fm.max(1).values, 0.9)),
            float((mseed[: (N // 64) * 64].view(-1, 64).max(1).values / fm.max(1).values).median()),
            float(torch.quantile(mseed[: (N // 64) * 64].view(-1, 64).max(1).values / fm.max(1).values, 0.9))))
    abls = (("1", "no selection (MFMA + vote + control + staging)"), ("8", "MFMAs only, tiles from an L2-hot set"), ("9", "MFMAs only, no tile loads")) if LISTS is not None else None
    for a, name in abls or (("5", "appends without their stores"), ("7", "no final ranking"), ("1", "no selection (MFMA + vote + control + staging)"), ("3", "MFMAs only (no vote)"), ("4", "MFMAs only, a barrier every other tile"), ("10", "MFMAs only, no barrier"), ("8", "MFMAs only, tiles from an L2-hot set"), ("9", "MFMAs only, no tile loads")):
        env = dict(os.environ, MELD_KNN16_ABLATION=a)
        subprocess.run([sys.executable, __file__, str(n)], env=env)
elif abl == "99":
    pass  # (set-up only: tools/sim_wg_schedule.py, tools/list_stats.py import this module for it)
else:
    seed = torch.load("/tmp/knn_seed.pt").cuda()
    run(seed, "ablation %s" % abl); run(seed, "ablation %s" % abl)
