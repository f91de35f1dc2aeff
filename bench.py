#!/usr/bin/env python
"""bench.py -- cells/sec through MELD.fit_transform (kNN + Chebyshev) on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on rank 0.
A "step" is one full ``MELD(knn=15, beta=60, chebyshev_order=30).fit_transform(X, labels)`` on the
synthetic cells of SURVEY.md section 8(d): graph build (distance GEMM + top-k on the matrix cores,
exact refinement, symmetrisation, anisotropy), lmax, 30 Chebyshev steps, densities back on the
host as a DataFrame.  X is resident in HBM (fp64) when the timed region starts; labels are a host
array of strings (they are factorised inside the timed region, as in the reference).

Workload: BASELINE.json's metric is quoted at 1M cells x 50 dims (configs[3]); it fits one GPU, so
N=1 runs exactly that.  With --gpus N > 1 the cells are row-sharded over the ranks (strong scaling:
the problem size stays 1M cells).

Extra objects on the JSON line: ``roofline`` (dominant kernel: the kNN distance/top-k kernel,
MFMA-bound), ``roofline_cheby`` (the HBM-bound CSR recurrence the north star's 70 % target refers
to), ``cpu_baseline`` (the oracle -- an in-repo scipy/sklearn restatement of the reference path --
timed on a bounded sample on this box's host cores), ``stages`` (per-stage seconds of one step).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA peak
PEAK_MFMA_F16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured-achievable)


def cheby_bytes_per_step(nnz, n, p):
    """Algorithmic HBM bytes of one recurrence step (SURVEY.md 8d / BASELINE.md): fp64 values +
    int32 columns, row pointers, degree vector, five [N,p] vector passes (gather T_{k-1} counted
    once, read T_{k-2}, write T_k, read r, write r)."""
    return 12 * nnz + 4 * (n + 1) + 8 * n + 40 * n * p


def synthetic_cells(n_cells, n_dims=50, seed=0, latent_dim=10, n_clusters=20):
    """The seeded workload of SURVEY.md section 8(d): 20 cluster centres ~ N(0, 4 I_10), points =
    centre + N(0, I_10), embedded in ``n_dims`` by a fixed random orthonormal map, plus N(0, 0.05^2)
    isotropic noise; labels ~ Bernoulli(expit(latent_0)).  Same generator (same stream of draws) as
    the oracle's, restated here so that the timed path never touches ``oracle/``
    (tests/test_host_api.py checks the two agree bit for bit)."""
    rng = np.random.default_rng(seed)
    latent_dim = min(latent_dim, n_dims)
    centres = rng.normal(0.0, 2.0, size=(n_clusters, latent_dim))
    assign = rng.integers(0, n_clusters, size=n_cells)
    latent = centres[assign] + rng.normal(0.0, 1.0, size=(n_cells, latent_dim))
    q, _ = np.linalg.qr(rng.normal(size=(n_dims, latent_dim)))
    X = latent @ q.T + rng.normal(0.0, 0.05, size=(n_cells, n_dims))
    p = 1.0 / (1.0 + np.exp(-latent[:, 0]))
    labels = np.where(rng.random(n_cells) < p, "expt", "ctrl")
    return np.ascontiguousarray(X, dtype=np.float64), labels


def cpu_baseline(sample_cells, dims, knn, beta, order, n_target, full_protocol=False):
    """Oracle (kind='port': the in-repo scipy/sklearn restatement of the reference path, oracle/meld_oracle.py)
    timed on this box's host cores, SURVEY.md section 8(d): the reference stack's own configuration -- sklearn
    ball_tree kNN with n_jobs=1 (graphtools' defaults) and single-threaded scipy CSR products -- on three
    sample sizes; the kNN is super-linear in N, so the exponent is fitted and the figure at the benchmark size
    is EXTRAPOLATED and labelled as such.  `value` is the measured rate of the largest sample."""
    from oracle import meld_oracle as mo

    sizes = [50_000, 100_000, 200_000] if full_protocol else [sample_cells // 4, sample_cells // 2, sample_cells]
    runs = []
    for n in sizes:
        X, labels = mo.synthetic_cells(n, n_dims=dims, seed=0)
        t0 = time.perf_counter()
        mo.fit_transform(X, labels, knn=knn, beta=beta, chebyshev_order=order, algorithm="ball_tree", n_jobs=1)
        runs.append((n, time.perf_counter() - t0))
    ln = np.log([r[0] for r in runs])
    lt = np.log([r[1] for r in runs])
    expo, _ = np.polyfit(ln, lt, 1)
    n_big, t_big = runs[-1]
    X, labels = mo.synthetic_cells(n_big, n_dims=dims, seed=0)
    t0 = time.perf_counter()
    mo.fit_transform(X, labels, knn=knn, beta=beta, chebyshev_order=order, algorithm="brute", n_jobs=-1)
    t_best = time.perf_counter() - t0
    return {
        "value": n_big / t_big,
        "unit": "cells/s",
        "cores": 1,
        "kind": "port",
        "sample": "N={} cells x {} dims, full fit_transform, sklearn ball_tree n_jobs=1 + scipy CSR (the reference "
        "stack's defaults); {:.1f} s".format(n_big, dims, t_big),
        "runs": [{"cells": n, "seconds": round(t, 2), "cells_per_s": n / t} for n, t in runs],
        # (descriptive only: the ball tree's cost grows between N^1.6 at these sizes and N^2.7 past 50k cells -- the two fits of earlier
        # rounds disagreed 44x at 1M cells, so NO figure is extrapolated to the benchmark size any more; the measured sizes stand alone)
        "fitted_exponent_over_these_runs": float(expo),
        "host_cores_available": os.cpu_count(),
        "measured_full_size": measured_full_size(n_target),
        "speedup_basis": "a GPU / CPU ratio at the benchmark size may only use `measured_full_size` (whole oracle on all host cores, "
        "brute-force kNN: the fastest CPU configuration measured, REPLAYED from the file named in it) or the runs above at THEIR "
        "sizes; nothing on this line is extrapolated",
        "recorded_full_protocol": recorded_full_protocol(),
        "best_effort": {
            "value": n_big / t_best,
            "cores": os.cpu_count(),
            "note": "same {}-cell sample with sklearn brute-force kNN on all host cores (n_jobs=-1); {:.1f} s".format(n_big, t_best),
        },
    }


def measured_full_size(n_cells):
    """The one MEASURED CPU number at the benchmark size: the whole oracle (brute-force kNN on all host cores) run by the round's
    evidence script (tools/parity_200k.py under tools/_profile_round.sh, 160 s at 1M cells -- too long for a default bench run),
    recorded with its commit in profiles/r04_cpu_full_size.json."""
    for name in ("r06_cpu_full_size.json", "r05_cpu_full_size.json", "r04_cpu_full_size.json"):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
        try:
            with open(path) as f:
                rec = json.load(f).get(str(int(n_cells)))
        except Exception:
            rec = None
        if rec:
            # (first key: this object was not measured in this run)
            return dict({"replayed_from": "profiles/{}@{}".format(name, rec.get("commit", "?"))}, **rec)
    return None


def recorded_full_protocol():
    """SURVEY 8d's own sample sizes (50k / 100k / 200k) take 27 minutes of host time -- the ball tree degrades sharply past
    50k cells in 50 dimensions -- so they are timed once per round (`bench.py --cpu-full`, tools/_profile_round.sh) and the
    record, with the commit it was taken at, rides along in every line; the default run times a quarter of those sizes."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_cpu_baseline_full.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except Exception:
        return None
    # the measured runs only (the record's own extrapolation to 1M cells is not carried: see fitted_exponent_over_these_runs)
    keep = {k: rec[k] for k in ("commit", "host", "protocol", "runs", "best_effort", "note") if k in rec}
    return dict({"replayed_from": "profiles/r03_cpu_baseline_full.json@{}".format(rec.get("commit", "?"))}, **keep)


def cpu_chebyshev_full_size(G, labels, beta, order):
    """The filter stage of the CPU oracle at the FULL benchmark size, on the device-built CSR: D2H of
    W, pygsp-style recurrence in scipy (single-threaded CSR x dense), checked against the device result."""
    from scipy import sparse

    from meld_amd import filter as mfilter
    from oracle import meld_oracle as mo

    import torch

    n = G.N
    W = sparse.csr_matrix((G.val.cpu().numpy(), G.col.cpu().numpy(), G.rowptr.cpu().numpy()), shape=(n, n))
    L = (sparse.diags(G.dw_dev.cpu().numpy(), 0) - W).tocsr()
    lmax = G.lmax
    c = mo.cheby_coeff(mo.filter_kernel_fn("heat", beta, 0, 1, lmax), lmax, order)
    _, ind = mo.sample_indicators(labels)
    perm = G.perm.cpu().numpy() if G.perm is not None else np.arange(n)
    s_int = np.ascontiguousarray(ind[perm])
    t0 = time.perf_counter()
    ref = mo.cheby_op(L, lmax, c, s_int)
    t_cpu = time.perf_counter() - t0
    out = mfilter.chebyshev_apply(G, torch.from_numpy(s_int).cuda(), c, lmax).cpu().numpy()
    return {
        "cells": n,
        "seconds": t_cpu,
        "cores": 1,
        "what": "oracle cheby_op (scipy CSR x dense, {} products) on the device-built W, measured at full size".format(order),
        "max_rel_diff_vs_device": float(np.abs(out - ref).max() / np.abs(ref).max()),
    }


def measured_traffic(kernel, n_cells, dims):
    """HBM-side bytes per launch from a separate rocprofv3 --pmc pass (profiles/pmc/traffic.json, written by
    tools/pmc_traffic.py from TCC_EA0_RDREQ / WRREQ as MI355X_MICROARCH.md prescribes), when one was taken for this
    kernel at this problem size; None otherwise (a bench run itself collects no counters)."""
    path = os.path.join(ROOT, "profiles", "pmc", "traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    rec = table.get("{}@{}x{}".format(kernel, n_cells, dims))
    return rec


def replayed_from(rec):
    """`traffic` is not measured by a bench run (counters need their own rocprofv3 --pmc pass): say where it comes from."""
    return "profiles/pmc/traffic.json@{}".format(rec.get("commit", "?")) if rec else None


def cheby_roofline(G, ev, order, p, N, d):
    """`roofline_cheby` object from the recorded `cheby_steps` spans of graph G (HIP events around the recurrence launches)."""
    t_ch = float(np.mean(ev["cheby_steps"])) * 1e-3
    steps = order - 1
    rows = G.info.get("rows_local", N)
    byts = cheby_bytes_per_step(G.nnz, rows, p)
    tiled = G.info.get("spmm") == "tiled"
    tr = measured_traffic("pt_step" if tiled else "cheby_step", N, d)
    return {
        "kernel": ("pt_step_kernel<P=2> (panel-tiled, symmetry-folded Laplacian recurrence, iterate staged in LDS)" if tiled else
                   "cheby_step_kernel<P=2> (fused CSR Laplacian recurrence)") + ", {} launches".format(steps),
        "bound": "hbm",
        "achieved": byts * steps / t_ch / 1e9,
        "peak": PEAK_HBM_GBS,
        "unit": "GB/s",
        "frac": byts * steps / t_ch / 1e9 / PEAK_HBM_GBS,
        "traffic": (tr or {}).get("bytes_per_launch"),
        "traffic_replayed_from": replayed_from(tr),
        "traffic_note": (tr or {}).get("note", "no PMC pass on record for this size (profiles/pmc/traffic.json)"),
        "traffic_commit": (tr or {}).get("commit"),
        "algorithmic": "{} B per launch (12*nnz + 4(N+1) + 8N + 40*N*p)".format(byts),
        "us_per_launch": 1e6 * t_ch / steps,
        "cells": int(N),
        "nnz_W": int(G.nnz),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)  # (the first call pays one-off costs; on a cold box the second one still can)
    ap.add_argument("--cells", type=int, default=1_000_000)
    ap.add_argument("--dims", type=int, default=50)
    ap.add_argument("--knn", type=int, default=15)
    ap.add_argument("--beta", type=float, default=60)
    ap.add_argument("--order", type=int, default=30)
    ap.add_argument("--cpu-sample", type=int, default=50000,
                    help="largest CPU-baseline sample (0 = skip); also run at 1/2 and 1/4 of it.  Default 50000: 12.5k / 25k / "
                         "50k cells, about 45 s of host time (the line also carries the round's recorded run of SURVEY 8d's "
                         "own 50k / 100k / 200k, which takes 27 minutes: --cpu-full)")
    ap.add_argument("--cpu-full", action="store_true", help="time SURVEY 8d's 50k / 100k / 200k samples (27 minutes of host time)")
    ap.add_argument("--no-host-input", action="store_true", help="skip the extra untimed-region passes with X on the host")
    ap.add_argument("--stages", action="store_true", help="extra untimed step with per-stage host timers")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra untimed passes of the default 1M run (C3's recurrence "
                                                            "roofline at 500k cells, the unpruned search)")
    ap.add_argument("--vfc", action="store_true", help="also time VertexFrequencyCluster (filter-bank method) on the benchmark graph "
                                                       "(BASELINE configs[4] on one GPU) and report `vfc` / `roofline_vfc`")
    ap.add_argument("--force-sharded", action="store_true", help="use the row-sharded driver even with one rank (testing)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import meld_amd
    from meld_amd import graph as mgraph

    N, d = args.cells, args.dims
    X_host, labels = synthetic_cells(N, n_dims=d, seed=0)
    X = torch.from_numpy(X_host).cuda()
    del X_host

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        op = meld_amd.MELD(knn=args.knn, beta=args.beta, chebyshev_order=args.order, verbose=0)
        if sharded:
            from meld_amd import distributed as mdist

            return op, mdist.fit_transform_sharded(op, X, labels)
        return op, op.fit_transform(X, labels)

    # (as timeit does: no cyclic-GC pass of the interpreter inside the timed region -- with scipy / sklearn / pandas imported a
    # generation-2 collection takes ~45 ms, and one landed in the 5th of 5 steps of a default run: 98.9 ms against 52.2-52.4.
    # Collected BEFORE the warm-up: the pass idles the GPU for ~0.1 s, and the first step after such a pause runs ~2 ms slower)
    import gc

    gc.collect()
    gc.disable()
    op = dens = None
    mgraph.record_events(True)  # (as in the timed loop: the first spans create the event pool)
    for _ in range(args.warmup):
        # (results held exactly as in the timed loop: the previous step's graph is alive while the next one is built, and a
        # warm-up that dropped it at once left the allocator two steps short of its steady state -- +1.8 ms on the first
        # two timed steps)
        op, dens = one_step()
    mgraph.record_events(True)
    barrier()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        op, dens = one_step()  # (synchronous: the densities come back as a host DataFrame)
        step_ms.append(1e3 * (time.perf_counter() - ts))
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    ev = mgraph.event_times_ms()
    mgraph.record_events(False)
    # the same step with X handed over as a HOST array (SURVEY 8d counts the H2D copy; `value` does not -- the bench contract
    # starts with the inputs resident): the SAME protocol as `value` -- one warm-up, then args.steps steps in one
    # synchronize-bracketed region
    host_ms, host_elapsed = [], None
    if world == 1 and not args.no_host_input:
        X_host = X.cpu().numpy()
        meld_amd.MELD(knn=args.knn, beta=args.beta, chebyshev_order=args.order, verbose=0).fit_transform(X_host, labels)
        gc.collect()
        gc.disable()
        torch.cuda.synchronize()
        th = time.perf_counter()
        for i in range(args.steps):
            ts = time.perf_counter()
            meld_amd.MELD(knn=args.knn, beta=args.beta, chebyshev_order=args.order, verbose=0).fit_transform(X_host, labels)
            host_ms.append(1e3 * (time.perf_counter() - ts))
        torch.cuda.synchronize()
        host_elapsed = time.perf_counter() - th
        gc.enable()
        del X_host
    per_rank = None
    if dist is not None:
        # what every rank saw (outside the timed region): its own wall time over the steps, the HIP-event spans of its stages, its
        # share of the graph -- so that the first run on a multi-GPU node says where a rank's time goes
        mine = {"rank": rank, "elapsed_s": elapsed, "ms_per_step": 1e3 * elapsed / args.steps,
                "events_ms": {k: float(np.mean(v)) for k, v in ev.items()},
                "rows_local": int(op.graph.info.get("rows_local", 0)), "nnz_local": int(op.graph.nnz),
                "exchange": op.graph.info.get("exchange"), "exchange_capacity": op.graph.info.get("exchange_capacity"),
                "two_phase": bool(op.graph.info.get("two_phase")), "principal_frame": bool(op.graph.info.get("principal_frame")),
                "rccl_c_loops": bool(getattr(getattr(op.graph, "comm", None), "rccl", lambda: None)() is not None)}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    G = op.graph
    nnz = int(G.info.get("nnz_global", G.nnz))
    p = dens.shape[1]
    G.estimate_lmax()
    out = {
        "metric": "cells/sec through MELD.fit_transform (kNN+Chebyshev)",
        "value": N * args.steps / elapsed,
        "unit": "cells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "ms_per_step_median": float(np.median(step_ms)),
        "ms_per_step_all": [round(t, 3) for t in step_ms],
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "{} cells x {} dims ({}), 20-cluster 10-d latent mixture, seed 0; `value`: X resident in HBM (fp64) "
            "when the timed region starts (`value_host_input`: the same step with X handed over as a host array, SURVEY "
            "8d's host-visible definition), labels host strings, densities back as a host DataFrame; "
            "knn={}, decay=40, thresh=1e-4, anisotropy=1, beta={}, heat filter, chebyshev_order={}, p={} labels".format(
                N, d, {1_000_000: "BASELINE configs[3] size, the size the metric is quoted on", 500_000: "BASELINE configs[2]",
                       50_000: "BASELINE configs[1]"}.get(N, "custom size"), args.knn, args.beta, args.order, p),
            "spmm_kernel": G.info.get("spmm"),
            "search_options": {k: G.info.get(k) for k in ("search", "nprod", "prune", "radius_cut", "seed", "step_lists", "principal_frame") if k in G.info},
            "parallelism": "single GPU" if world == 1 else "rows sharded over {} GPUs, all-gather per Chebyshev step".format(world),
            "nnz_W": nnz,
            "mean_degree": nnz / N,
            "rows_through_exact_sweep": int(G.info.get("n_flagged_rows", 0)),
            "rows_researched_full_precision": int(G.info.get("n_researched_rows", 0)),
            "lmax": float(G.lmax),
            "lanczos_iterations": int(G.lmax_info.get("iterations", 0)),
        },
    }
    if per_rank is not None:
        # per Chebyshev step every rank all-gathers the iterate (8 p bytes per cell), per Lanczos iteration the iterate (8 bytes per
        # cell) and 3 x 64 doubles; the symmetrisation is one fixed-capacity all-to-all
        out["per_rank"] = per_rank
        out["collective_bytes_per_step"] = {
            "chebyshev_all_gather": int((args.order - 1) * 8 * p * N), "lanczos_all_gather": int(G.lmax_info.get("iterations", 0)) * 8 * N,
            "symmetrise_all_to_all_capacity": int(world * 16 * int(G.info.get("exchange_capacity") or 0)), "kernel_row_sums_all_gather": 8 * N}
    if "knn_topk" in ev:
        t_knn = float(np.mean(ev["knn_topk"])) * 1e-3
        rows = G.info.get("rows_local", N)
        flops = 2.0 * rows * N * d
        search = G.info.get("search", "f16x3")
        computed_frac = 1.0
        if search == "f16x3":
            kb = (d + 15) // 16
            kp = 16 * kb
            nprod = int(G.info.get("nprod", 3))
            # 16-deep K blocks issued per pair: x3 with the full split (hi.hi + hi.lo + lo.hi), x1 on the
            # hi parts alone (the norms are added outside the MFMAs)
            blocks = 3 * kb if nprod == 3 else kb
            executed = 2.0 * rows * N * 16 * blocks
            # exact tile pruning: the kernel counts the (64 queries x 64 references) blocks it really computed
            wt = G.info.get("wave_tiles_done")
            partial = None
            if wt:
                executed = 2.0 * 64 * 64 * 16 * blocks * float(wt)
                computed_frac = float(wt) / (math.ceil(rows / 64) * math.ceil(N / 64))
                on = G.info.get("blocks_past_partial_test")
                kept = G.info.get("pairs_past_filter")  # two-pass route (round 6): (wave, tile) pairs the list-filter pass left to the search
                if on is not None and nprod == 1:
                    # partial test (principal frame): every (wave, tile) pair the search kernel computes issues K block 0 of its two
                    # blocks of 32 references; only the blocks that pass the test issue the other kb - 1 K blocks.  Two-pass route:
                    # the pairs the search computes are the ones the filter pass kept; the filter pass's own flops (K block 0 of
                    # every listed pair) are reported under `roofline_filter`
                    searched = float(kept) if kept is not None else float(wt)
                    executed = 2.0 * 16 * (64 * 64 * searched + 32 * 64 * (kb - 1) * float(on))
                    partial = {"blocks_of_32_refs_past_the_first_k_block": int(on), "of": int(2 * searched), "frac": float(on) / (2.0 * searched)}
                    if kept is not None:
                        partial.update(pairs_listed_and_tested_by_the_filter_pass=int(wt), pairs_it_kept=int(kept), kept_frac=float(kept) / float(wt))
            peak, kname = PEAK_MFMA_F16_TFLOPS, "knn16_topk_kernel (split-fp16 hi/lo distance GEMM on v_mfma_f32_32x32x16_f16 + streaming top-k)"
        else:
            kp = int(G.info.get("KP", d + 2))
            executed = 2.0 * rows * N * kp
            peak, kname = PEAK_MFMA_F32_TFLOPS, "knn_topk_kernel (fp32 MFMA distance GEMM + streaming top-k)"
        out["roofline"] = {
            "kernel": kname,
            "bound": "mfma",
            # achieved / frac = flops ISSUED to the matrix pipe per kernel time (the kernel's own count of computed
            # 64 x 64 blocks x K padded to a multiple of 16): a utilisation.  The brute-force-equivalent rate
            # 2 N^2 d / t (SURVEY 8d's algorithmic figure) is kept beside it: with exact pruning it says how much
            # faster than the specified GEMM the kernel is, not how busy the pipe is.
            "achieved": executed / t_knn / 1e12,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": executed / t_knn / 1e12 / peak,
            "traffic": (measured_traffic("knn16_topk", N, d) or {}).get("bytes_per_launch"),
            "traffic_replayed_from": replayed_from(measured_traffic("knn16_topk", N, d)),
            # the replayed traffic over THIS run's kernel time: what the fabric side of the L2s delivers (a plain copy reaches ~6300 GB/s)
            "traffic_rate_gb_s": ((measured_traffic("knn16_topk", N, d) or {}).get("bytes_per_launch") or 0.0) / t_knn / 1e9 or None,
            "bound_note": "`frac` = flops issued to the matrix pipe / dense f16 peak (a utilisation).  Round 6: the partial test runs as a pass "
            "of its own over the step lists (`roofline_filter`), this kernel stages whole tiles only for the pairs that pass kept (a quarter) -- "
            "its tile stream (`traffic`, replayed from the PMC pass; `traffic_rate_gb_s` over this run's kernel time) is no longer at the "
            "fabric's copy rate (~6300 GB/s); what it runs into is its selection slow path (two thirds of its blocks hold a candidate)",
            "traffic_note": (measured_traffic("knn16_topk", N, d) or {}).get(
                "note", "no PMC pass on record for this size (profiles/pmc/traffic.json); a bench run collects no counters"),
            "traffic_commit": (measured_traffic("knn16_topk", N, d) or {}).get("commit"),
            "algorithmic": "2*Nq*N*d = {:.3e} flop per launch".format(flops),
            "algorithmic_equiv_tflops": flops / t_knn / 1e12,
            "algorithmic_equiv_frac_of_peak": flops / t_knn / 1e12 / peak,
            "note": "executed = flops issued to the matrix pipe (K padded to {}{}); algorithmic rate is {:.2f}x the "
            "157.3 TF fp32-MFMA peak; first-pass kernel only, the re-search of {} uncertified rows is reported under "
            "stages".format(kp, ", split-fp16 products nprod={}".format(G.info.get("nprod")) if search == "f16x3" else "",
                            flops / t_knn / 1e12 / PEAK_MFMA_F32_TFLOPS, G.info.get("n_researched_rows", 0)),
            "blocks_computed_frac": computed_frac,
            "partial_test": (partial if search == "f16x3" else None),
            "pruning_note": "algorithmic = the brute-force distance GEMM the path is specified by (SURVEY.md 8d); exact "
            "tile pruning (triangle-inequality bounds, results unchanged) lets a wave skip the (64 x 64) blocks that "
            "cannot hold a neighbour: blocks_computed_frac of them are computed, `executed` counts only those; in the cells' "
            "principal frame a computed block of 32 references is tested behind its first K block (a distance over some coordinates "
            "never exceeds the distance) and issues its other K blocks only if some partial value is within reach of its row: "
            "`partial_test` counts those, `executed` = K block 0 of every computed block + the other K blocks of the ones that went on",
            "ms": 1e3 * t_knn,
        }
    if "knn_filter" in ev and "roofline" in out and G.info.get("two_phase"):
        # the list-filter pass in front of the search (meld_knn16_partial_filter): K block 0 of every listed (wave, tile) pair, 4 MFMAs
        t_f = float(np.mean(ev["knn_filter"])) * 1e-3
        wt = float(G.info.get("wave_tiles_done") or 0)
        ex_f = 2.0 * 64 * 64 * 16 * wt
        out["roofline_filter"] = {
            "kernel": "knn16_partial_filter_kernel<8, 4> (K block 0 of every listed (wave, tile) pair on v_mfma_f32_32x32x16_f16, lists thinned in place)",
            "bound": "mfma", "achieved": ex_f / t_f / 1e12, "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s", "frac": ex_f / t_f / 1e12 / PEAK_MFMA_F16_TFLOPS,
            "ms": 1e3 * t_f, "pairs_tested": int(wt), "pairs_kept": G.info.get("pairs_past_filter"),
            "traffic": (measured_traffic("knn16_partial_filter", N, d) or {}).get("bytes_per_launch"),
            "traffic_replayed_from": replayed_from(measured_traffic("knn16_partial_filter", N, d)),
            "note": "HIP events around the launch (+ the re-sort of the dispatch order behind it); issue-bound: one LDS-DMA piece costs ~180 "
                    "cycles of issue, the test ~300 per pair (4 MFMAs = 128 pipe cycles + the minima over 64 accumulator registers); "
                    "without its staging the pass takes 5.2 ms, without its arithmetic 2.7, with neither 1.0 (DESIGN.md 4.1.8)",
        }
        t_both = t_f + float(np.mean(ev["knn_topk"])) * 1e-3
        out["roofline"]["search_ms_filter_plus_search"] = 1e3 * t_both
        # the two launches of the search stage together: flops issued by both over the time of both (a utilisation, like `frac`)
        out["roofline"]["frac_filter_plus_search"] = (ex_f + out["roofline"]["achieved"] * 1e12 * float(np.mean(ev["knn_topk"])) * 1e-3) / t_both / 1e12 / PEAK_MFMA_F16_TFLOPS
    if "cheby_steps" in ev:
        out["roofline_cheby"] = cheby_roofline(G, ev, args.order, p, N, d)
    if world == 1 and not args.no_extra and N == 1_000_000 and d == 50:
        # BASELINE configs[2] (C3, 500k x 50) is the configuration the north star's 70 % recurrence target is quoted on: one extra
        # untimed pass at that size (a warm-up + two steps), its recurrence launches timed by the same HIP events
        Xc, lc = synthetic_cells(500_000, n_dims=d, seed=0)
        Xc = torch.from_numpy(Xc).cuda()
        opc = None
        for i in range(3):
            if i == 1:
                torch.cuda.synchronize()
                mgraph.record_events(True)
            opc = meld_amd.MELD(knn=args.knn, beta=args.beta, chebyshev_order=args.order, verbose=0)
            tc = time.perf_counter()
            densc = opc.fit_transform(Xc, lc)
            tc = time.perf_counter() - tc
        torch.cuda.synchronize()
        evc = mgraph.event_times_ms()
        mgraph.record_events(False)
        if "cheby_steps" in evc:
            out["roofline_cheby_c3"] = dict(cheby_roofline(opc.graph, evc, args.order, densc.shape[1], 500_000, d),
                                            ms_per_step_whole_fit_transform=1e3 * tc,
                                            note="BASELINE configs[2] (500k x 50): untimed extra pass of this run, X resident")
        del Xc, opc, densc
        # the reference's own default width: graphtools' n_pca = 100 (forwarded at reference meld/meld.py:117-118) hands the graph
        # builder 100 principal components -- one extra untimed pass over 1M x 100 cells of the same mixture (a warm-up + one step):
        # seven K blocks, the principal frame by the library rotation, the same two search passes
        try:
            Xw, lw = synthetic_cells(N, n_dims=100, seed=0)
            Xw = torch.from_numpy(Xw).cuda()
            opw = None
            for i in range(2):
                if i == 1:
                    torch.cuda.synchronize()
                    mgraph.record_events(True)
                opw = meld_amd.MELD(knn=args.knn, beta=args.beta, chebyshev_order=args.order, verbose=0)
                tw = time.perf_counter()
                opw.fit_transform(Xw, lw)
                tw = time.perf_counter() - tw
            torch.cuda.synchronize()
            evw = mgraph.event_times_ms()
            mgraph.record_events(False)
            iw = opw.graph.info
            out["roofline_d100"] = {
                "workload": "1000000 cells x 100 dims (the reference's default n_pca), same mixture, seed 0",
                "ms_per_step_whole_fit_transform": 1e3 * tw, "cells_per_s_whole_fit_transform": N / tw,
                "search_filter_ms": float(np.mean(evw["knn_filter"])) if "knn_filter" in evw else None,
                "search_ms": float(np.mean(evw["knn_topk"])) if "knn_topk" in evw else None,
                "principal_frame": bool(iw.get("principal_frame")), "two_phase": bool(iw.get("two_phase")),
                "pairs_listed": iw.get("wave_tiles_done"), "pairs_past_filter": iw.get("pairs_past_filter"),
                "blocks_past_partial_test": iw.get("blocks_past_partial_test"), "nnz_W": int(opw.graph.nnz),
                "note": "untimed extra pass of this run (a warm-up + one step), X resident; HIP events around the two search launches",
            }
            del Xw, opw
        except Exception as e:  # (an extra: never the reason a bench line is missing)
            out["roofline_d100"] = {"error": repr(e)}
            mgraph.record_events(False)
        # the data dependence of the search on the line: the same step with the exact tile pruning switched off (every
        # 64 x 64 block computed -- what iid data without cluster structure would cost), one untimed step
        os.environ["MELD_KNN_PRUNE"] = "0"
        dev_was = os.environ.get("MELD_DEV")
        os.environ["MELD_DEV"] = "1"  # (a development switch: read only under MELD_DEV=1, meld_amd/_options.py)
        opu = None
        try:
            mgraph.record_events(True)
            opu = meld_amd.MELD(knn=args.knn, beta=args.beta, chebyshev_order=args.order, verbose=0)
            tu = time.perf_counter()
            opu.fit_transform(X, labels)
            tu = time.perf_counter() - tu
            torch.cuda.synchronize()
            evu = mgraph.event_times_ms()
        finally:
            del os.environ["MELD_KNN_PRUNE"]
            if dev_was is None:
                del os.environ["MELD_DEV"]
            else:
                os.environ["MELD_DEV"] = dev_was
            mgraph.record_events(False)
        if "knn_topk" in evu and "roofline" in out:
            t_u = float(np.mean(evu["knn_topk"])) * 1e-3
            kbu = (d + 3 + 15) // 16
            out["roofline"]["unpruned"] = {
                "ms": 1e3 * t_u, "ms_per_step_whole_fit_transform": 1e3 * tu, "cells_per_s_whole_fit_transform": N / tu,
                "achieved": 2.0 * N * N * 16 * kbu / t_u / 1e12, "frac": 2.0 * N * N * 16 * kbu / t_u / 1e12 / PEAK_MFMA_F16_TFLOPS,
                "note": "MELD_KNN_PRUNE=0: same kernel, same graph, every 64 x 64 block computed (no tile bounds, no step lists); "
                        "the pruned rate above depends on the cluster structure of the data, this one does not",
            }
        del opu
    if args.stages and world == 1:
        op2 = meld_amd.MELD(knn=args.knn, beta=args.beta, chebyshev_order=args.order)
        t0 = time.perf_counter()
        op2.fit(X, profile=True)
        torch.cuda.synchronize()
        t_fit = time.perf_counter() - t0
        t0 = time.perf_counter()
        op2.transform(labels)
        torch.cuda.synchronize()
        out["stages"] = dict(op2.graph.info["stage_seconds"], fit_total=t_fit, transform_total=time.perf_counter() - t0)
    if args.vfc and world == 1:
        # BASELINE configs[4] on one GPU: the filter-bank VertexFrequencyCluster on the graph just built -- n_probes
        # columns through the same recurrence kernel (2 M + 1 SpMMs), Ritz pairs + band energies, PCA + KMeans
        lik = meld_amd.utils.normalize_densities(dens)
        mgraph.record_events(True)
        t0 = time.perf_counter()
        vfc = meld_amd.VertexFrequencyCluster(n_clusters=6, random_state=0, n_init=2)
        vfc.fit(G)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        vfc.transform(op.sample_indicators["expt"], lik["expt"])
        t2 = time.perf_counter()
        clusters = vfc.predict()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ev2 = mgraph.event_times_ms()
        mgraph.record_events(False)
        out["vfc"] = {
            "method": vfc.method_, "cells": N, "n_probes": int(vfc.n_probes), "n_bands": int(vfc.n_bands), "windows": len(vfc.window_sizes),
            "chebyshev_order": int(vfc._fb["order"]), "fit_s": t1 - t0, "transform_s": t2 - t1, "predict_s": t3 - t2,
            "clusters": int(len(np.unique(clusters))),
            "note": "filter-bank method (a new algorithm: the reference's dense one needs an N x N Fourier basis); fit = "
                    "2 M + 1 SpMMs of n_probes columns on the recurrence kernel + QR / Rayleigh-Ritz (library)",
        }
        if "vfc_spmm" in ev2:
            R = int(vfc.n_probes)
            t_sp = float(np.median(ev2["vfc_spmm"])) * 1e-3  # (median: the first product of a fit also pays one-off costs)
            byts = cheby_bytes_per_step(G.nnz, N, R)
            wide = getattr(vfc, "_fb", {}).get("spmm") == "wide"
            out["roofline_vfc"] = {
                "kernel": ("cheby_step_wide_kernel: lanes = columns, one launch per SpMM of {} columns (the matrix is streamed once)".format(R) if wide else
                           "pt_step_kernel<P=2>, {} launches per SpMM of {} columns (the matrix is streamed once per column pair)".format((R + 1) // 2, R)),
                "bound": "hbm", "achieved": byts / t_sp / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": byts / t_sp / 1e9 / PEAK_HBM_GBS,
                "algorithmic": "{} B per SpMM (12*nnz + 4(N+1) + 8N + 40*N*p, p = {}: the matrix bytes counted once)".format(byts, R),
                "ms_per_spmm": 1e3 * t_sp, "ms_per_spmm_mean": float(np.mean(ev2["vfc_spmm"])), "spmms": len(ev2["vfc_spmm"]), "traffic": None,
            }
    out["value_definition"] = ("X resident in HBM when the timed region starts (the bench contract: inputs resident, the PCIe-inclusive "
                               "rate is reported beside it as `value_host_input` and is SURVEY 8d's host-visible figure)")
    if host_ms:
        # SURVEY 8d's metric counts the H2D copy of X: this is the figure to compare with it (`value` starts X-resident)
        out["value_host_input"] = N * len(host_ms) / host_elapsed
        out["host_input"] = {
            "ms_per_step": 1e3 * host_elapsed / len(host_ms),
            "ms_per_step_median": float(np.median(host_ms)),
            "steps": len(host_ms),
            "value": N * len(host_ms) / host_elapsed,
            "unit": "cells/s",
            "note": "same step with X as a host ndarray (the {:.0f} MB H2D copy over PCIe inside the step), {} steps in one "
            "synchronize-bracketed region after one warm-up -- the protocol of `value`; SURVEY 8d's host-visible definition, "
            "reported beside `value`, which starts with X resident as the bench contract prescribes".format(N * d * 8 / 1e6, len(host_ms)),
        }
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample, d, args.knn, args.beta, args.order, N, full_protocol=args.cpu_full)
        out["cpu_baseline"]["chebyshev_at_full_size"] = cpu_chebyshev_full_size(G, labels, args.beta, args.order)
        ch = out["cpu_baseline"]["chebyshev_at_full_size"]
        if "cheby_steps" in ev:
            ch["device_seconds_same_stage"] = float(np.mean(ev["cheby_steps"])) * 1e-3 * args.order / max(args.order - 1, 1)
    if dist is not None:
        dist.barrier()
    if rank == 0:
        # RCCL prints its version banner through C stdio at communicator set-up, which (piped) would only be
        # flushed at exit -- after Python's output; flush it now so that the JSON line is the last line on stdout.
        # (Printed before the process group is torn down: a rank that is slow to leave cannot hold the result back.)
        import ctypes

        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
