"""Time the recurrence step on a saved graph (tools/save_graph.py) with whatever library MELD_HIP_LIB names:
python tools/spmm_time.py graph.pt [reps]   -> one line per kernel; results checked against the CSR-stream kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
from meld_amd.graph import DeviceGraph, HipOps
from bench import cheby_bytes_per_step

d = torch.load(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
if os.environ.get("PT_DROP_LOWER_INBLOCK", "0") != "0":
    # timing-only: what would the stream cost with the in-block lower triangle gone (symmetric in-block storage)?
    R = int(os.environ["PT_DROP_LOWER_INBLOCK"])
    rp, col, val = d["rowptr"].cuda(), d["col"].cuda(), d["val"].cuda()
    rows = torch.repeat_interleave(torch.arange(rp.numel() - 1, device="cuda"), rp[1:] - rp[:-1])
    keep = ~((rows // R == col.to(torch.int64) // R) & (col.to(torch.int64) < rows))
    cnt = torch.zeros(rp.numel() - 1, dtype=torch.int64, device="cuda").index_add_(0, rows[keep], torch.ones_like(rows[keep]))
    rp2 = torch.zeros_like(rp); rp2[1:] = torch.cumsum(cnt, 0)
    print("dropped %.3f of the entries" % (1 - float(keep.double().mean())))
    d = dict(rowptr=rp2.cpu(), col=col[keep].cpu(), val=val[keep].cpu(), dw=d["dw"], N=d["N"])
G = DeviceGraph(d["rowptr"].cuda(), d["col"].cuda(), d["val"].cuda(), d["dw"].cuda())
n = G.N
tag = os.path.basename(os.environ.get("MELD_HIP_LIB", "base"))


def timed(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e3)
    return min(ts), float(np.median(ts))


from meld_amd._lib import get_lib
res = {}
MASK = int(os.environ.get("PT_MASK", "0"))  # timing-only ablation mask of meld_pt_debug_ablate (results wrong)
for mode in ("csr", "tiled"):
    G.pt = None
    G.ops = HipOps(spmm=mode)
    G.ops.pt_layout(G)
    for pp in (2, 1):
        gen = torch.Generator(device="cuda").manual_seed(pp)
        x = torch.rand(n, pp, dtype=torch.float64, device="cuda", generator=gen)
        z = torch.rand(n, pp, dtype=torch.float64, device="cuda", generator=gen)
        y = torch.empty_like(x)
        r = torch.zeros_like(x)
        G.ops.cheby_step(G, pp, x, 0, z, y, r, 0.7, -0.2, -1.0, 0.1)
        res[(mode, pp)] = y.cpu().numpy().copy()
        if mode == "tiled":
            if MASK:
                from meld_amd._lib import get_lib
                get_lib().meld_pt_debug_ablate(MASK)
                tag = tag.split(" mask")[0] + " mask%d" % MASK
            # ping-pong like the filter does (x and y swap every step)
            bufs = [x, y]
            state = {"i": 0}
            def step():
                i = state["i"]
                G.ops.cheby_step(G, pp, bufs[i & 1], 0, bufs[(i + 1) & 1], bufs[(i + 1) & 1], r, 1e-3, 0.5, 0.5, 0.0)
                state["i"] = i + 1
            best, med = timed(step, reps)
            byts = cheby_bytes_per_step(int(os.environ.get('PT_NNZ_FULL', G.nnz)), n, pp)
            err = np.abs(res[("tiled", pp)] - res[("csr", pp)]).max() / np.abs(res[("csr", pp)]).max()
            print("%-28s %s p=%d: best %.1f us median %.1f us  frac(best) %.3f  max rel diff vs csr %.1e"
                  % (tag, G.info.get("spmm"), pp, best, med, byts / best / 1e3 / 8000, err), flush=True)

# the Lanczos SpMV of the lmax estimate (p = 1, fp32 copy of the values, scalars from device memory)
state = torch.zeros(8, dtype=torch.float64, device="cuda"); state[3], state[4] = 0.5, -0.25
dots = torch.zeros(2 * G.ops.dot_slots(), dtype=torch.float64, device="cuda")
x1 = torch.rand(n, dtype=torch.float64, device="cuda"); z1 = torch.rand(n, dtype=torch.float64, device="cuda"); y1 = torch.empty_like(x1)
best, med = timed(lambda: G.ops.lanczos_spmv(G, x1, z1, y1, state, dots), reps)
print("%-28s lanczos spmv (p=1, fp32 values): best %.1f us median %.1f us" % (tag, best, med), flush=True)


# a plain streaming copy of the step's byte count beside it (the yardstick for "what does a byte cost": the PMC passes of
# tools/pmc_spmm.sh print its counters next to pt_step's): half the bytes read, half written, one elementwise torch kernel (x * c)
byts2 = cheby_bytes_per_step(G.nnz, n, 2)
src = torch.empty(byts2 // 16, dtype=torch.float64, device="cuda").normal_()
dst = torch.empty_like(src)
best, med = timed(lambda: torch.mul(src, 1.0000001, out=dst), reps)  # (an elementwise kernel with a name of its own in the trace)
print("%-28s plain copy of the p = 2 step's bytes (%d B read + written): best %.1f us median %.1f us  frac(best) %.3f"
      % (tag, 2 * src.numel() * 8, best, med, 2 * src.numel() * 8 / best / 1e3 / 8000), flush=True)
