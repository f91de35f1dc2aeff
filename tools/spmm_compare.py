"""CSR-stream vs panel-tiled recurrence kernel on the benchmark graph: python tools/spmm_compare.py [N] [p]
Prints the layout build time, us per Chebyshev step and algorithmic GB/s (SURVEY 8d bytes) for both kernels,
the Lanczos SpMV (p = 1) too, and the maximum difference of the results."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
import meld_amd
from meld_amd import graph as mg
from meld_amd.graph import HipOps
from bench import synthetic_cells, cheby_bytes_per_step

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 2
X, labels = synthetic_cells(n, 50, seed=0)
G = meld_amd.build_knn_graph(torch.from_numpy(X).cuda(), knn=15)
print("N=%d nnz=%d" % (n, G.nnz), flush=True)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us


res = {}
if os.environ.get("PT_FORCE_ABLATE", "0") != "0":  # profile an ablated variant (tools/pmc_spmm.sh)
    from meld_amd._lib import get_lib
    get_lib().meld_pt_debug_ablate(int(os.environ["PT_FORCE_ABLATE"]))
for mode in ("csr", "tiled"):
    G.pt = None
    G.ops = HipOps(spmm=mode)
    t0 = time.perf_counter()
    G.ops.pt_layout(G)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    if mode == "tiled":
        G.pt = None
        t0 = time.perf_counter()
        G.ops.pt_layout(G)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        print("  layout:", G.info.get("spmm"), flush=True)
        nt = G.pt["tensors"]["blk_ntile"].cpu().numpy()
        nd = G.pt["tensors"]["blk_ndist"].cpu().numpy()
        br = G.pt["tensors"]["blk_row"].cpu().numpy()
        rp = G.rowptr.cpu().numpy()
        ent = np.diff(rp[br])
        print("  layout: nb=%d build %.2f ms; tiles/block mean %.1f max %d; distinct cols/block mean %.0f; rows/block %d..%d; entries/block %d..%d"
              % (G.pt["nb"], 1e3 * t_build, nt.mean(), nt.max(), nd.mean(), np.diff(br).min(), np.diff(br).max(), ent.min(), ent.max()), flush=True)
    if mode == "tiled" and os.environ.get("PT_BUILD_STAGES", "0") != "0":
        from meld_amd._lib import get_lib
        for stage in (1, 2, 3, 0):
            get_lib().meld_pt_debug_ablate(stage << 8)
            G.pt = None
            us = timed(lambda: (setattr(G, "pt", None), G.ops.pt_layout(G)), 3)
            print("  layout build stopped after stage %d: %.2f ms" % (stage, us / 1e3), flush=True)
        get_lib().meld_pt_debug_ablate(0)
        G.pt = None
        G.ops.pt_layout(G)
    for pp in (p, 1):
        gen = torch.Generator(device="cuda").manual_seed(pp)
        x = torch.rand(n, pp, dtype=torch.float64, device="cuda", generator=gen)
        z = torch.rand(n, pp, dtype=torch.float64, device="cuda", generator=gen)
        y = torch.empty_like(x)
        r = torch.zeros_like(x)
        us = timed(lambda: G.ops.cheby_step(G, pp, x, 0, z, y, r, 0.7, -0.2, -1.0, 0.1), 20)
        byts = cheby_bytes_per_step(G.nnz, n, pp)
        print("  %-5s p=%d: %.1f us/step  %.0f GB/s algorithmic = %.3f of 8 TB/s" % (mode, pp, us, byts / us / 1e3, byts / us / 1e3 / 8000), flush=True)
        res[(mode, pp)] = y.cpu().numpy().copy()
if os.environ.get("PT_ABLATE", "1") != "0":
    from meld_amd._lib import get_lib
    lib = get_lib()
    for mask, what in ((4, "no panel loads"), (8, "panel gathers from a 16 KB window"), (16, "consumers ignore panel readiness")):
        lib.meld_pt_debug_ablate(mask)
        for pp in (p, 1):
            x = torch.rand(n, pp, dtype=torch.float64, device="cuda")
            y = torch.empty_like(x)
            us = timed(lambda: G.ops.cheby_step(G, pp, x, 0, x, y, None, 0.7, -0.2, -1.0, 0.1), 20)
            print("  ablation %d (%s) p=%d: %.1f us/step" % (mask, what, pp, us), flush=True)
    lib.meld_pt_debug_ablate(0)
for pp in (p, 1):
    a, b = res[("csr", pp)], res[("tiled", pp)]
    print("  max |tiled - csr| / max|csr| (p=%d): %.2e" % (pp, np.abs(a - b).max() / np.abs(b).max()))
