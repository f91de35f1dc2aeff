"""Per-rank compute of the row-sharded driver, measured on ONE GPU: rank r of a pretended world of G ranks runs the real
`fit_transform_sharded` with a stand-in Comm whose collectives move no data between ranks (all-gather: the local block is
copied into every slot; all-reduce: nothing; fixed-capacity exchange: the send
buffer comes back as the receive buffer).  The RESULTS ARE WRONG -- only the shapes, the launches and the work per rank are
those of a real rank -- so this is a timing tool: it tells what a rank computes per step, to which the cost of the
collectives has to be added.
python tools/shard_emulate.py [N] [world] [rank]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
import meld_amd
from meld_amd import distributed as mdist, graph as mg
from bench import synthetic_cells

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
WORLD = int(sys.argv[2]) if len(sys.argv) > 2 else 8
RANK = int(sys.argv[3]) if len(sys.argv) > 3 else 3


class FakeComm:
    group = None

    def __init__(self, world, rank):
        self.world, self.rank = world, rank
        self.calls = {}

    def _count(self, name, nbytes):
        c = self.calls.setdefault(name, [0, 0])
        c[0] += 1
        c[1] += int(nbytes)

    def all_gather_rows(self, full, local):
        self._count("all_gather", full.numel() * full.element_size())
        n = local.shape[0]
        for r in range(self.world):  # (every slot gets the local block: values that are used as indices stay in range)
            full[r * n : (r + 1) * n].copy_(local)

    def all_reduce_sum(self, t):
        self._count("all_reduce", t.numel() * t.element_size())
        return t

    def all_reduce_max(self, t):
        self._count("all_reduce", t.numel() * t.element_size())
        return t

    def exchange_fixed(self, send, cap):
        self._count("all_to_all", send.numel() * send.element_size())
        r = send.view(self.world, 2, cap)
        return r[:, 0, :].reshape(-1), r[:, 1, :].reshape(-1).view(torch.float64)

    _handle = None

    def rccl(self):
        """RCCL=1 (rank 0 only: its slice starts at row 0): a REAL one-rank RCCL communicator of the library, so that the
        recurrences take the C-side loops (meld_cheby_run_sharded / meld_lanczos_steps_sharded) -- kernel + ncclAllGather
        (+ ncclAllReduce) enqueued per step from one call; the one-rank collectives move nothing, like the stand-ins above,
        but cost their real issue time."""
        if os.environ.get("RCCL") != "1" or self.rank != 0:
            return None
        if FakeComm._handle is None:
            import ctypes as C
            from meld_amd._lib import check, get_lib
            lib = get_lib()
            buf = C.create_string_buffer(128)
            check(lib.meld_rccl_unique_id(buf), "meld_rccl_unique_id")
            h = C.c_void_p()
            check(lib.meld_rccl_comm_create(buf.raw, 1, 0, C.byref(h)), "meld_rccl_comm_create")
            FakeComm._handle = h
        return FakeComm._handle


if os.environ.get("TRACE"):  # name the stage a fault happens in: synchronise and print after every ops call
    from meld_amd.graph import HipOps

    def traced(name, fn):
        def w(*a, **k):
            r = fn(*a, **k)
            torch.cuda.synchronize()
            print("  done", name, flush=True)
            return r
        return w
    for name in ("directed_kernel_coo", "partition_remote", "assemble_rows", "row_sums", "anisotropy", "pt_layout", "lanczos_spmv",
                 "lanczos_fold", "lanczos_axpy3", "cheby_step", "sort_pairs"):
        setattr(HipOps, name, traced(name, getattr(HipOps, name)))

# three places where wrong data would change the WORK: the ordering (garbage children = no locality = no pruning) is computed
# unsharded (its assignment passes then cost 8/8 instead of 1/8 of 0.85 ms), and the lmax estimate (garbage vectors never
# converge) is stopped after the 35 iterations the real graph needs
from meld_amd import reorder as _ro, filter as _mf
_lp = _ro.locality_permutation
_ro.locality_permutation = lambda X, *a, comm=None, **k: _lp(X, *a, **k)
from meld_amd.graph import HipOps as _HipOps
_HipOps.shards_spheres = False  # (likewise: stand-in spheres of the other ranks' tiles would wreck the pruning; a real rank computes 1 / world of the 0.7 ms)
_fold = _mf._lanczos_lmax_folded
_mf._lanczos_lmax_folded = lambda G, ops, comm, u0, tol, max_iter, check_every: _fold(G, ops, comm, u0, tol, 35, check_every)
if os.environ.get("RCCL") == "1":
    print("recurrences through the C-side loops on a one-rank RCCL communicator (RCCL=1)")

from meld_amd.graph import HipOps as _HO
_pr = _HO.partition_remote
def _pr_print(self, keys, vals, R, world, rank, cap):
    send, counts = _pr(self, keys, vals, R, world, rank, cap)
    print("  entries owed to each rank", counts.tolist(), "of", int(keys.shape[0]), "capacity", cap)
    return send, counts
_HO.partition_remote = _pr_print

X, labels = synthetic_cells(N, 50, seed=0)
Xd = torch.from_numpy(X).cuda()
for rep in range(4):
    comm = FakeComm(WORLD, RANK)
    op = meld_amd.MELD(knn=15, beta=60, chebyshev_order=30, verbose=0)
    mg.record_events(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        mdist.fit_transform_sharded(op, Xd, labels, comm=comm)
    except Exception as e:  # (wrong data may trip a downstream check: the timing up to there still counts)
        print("stopped:", type(e).__name__, str(e)[:200])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ev = mg.event_times_ms()
    mg.record_events(False)
    print("rank %d of %d, N=%d: step %.2f ms (host wall, collectives free)  events %s" % (RANK, WORLD, N, 1e3 * (t1 - t0), {k: round(sum(v), 2) for k, v in ev.items()}))
print("collectives per step:", {k: (v[0], "%.1f MB" % (v[1] / 1e6)) for k, v in comm.calls.items()})
G = op.graph
print("lmax iterations", G.lmax_info.get("iterations"), "rows local", G.n_rows, "nnz local", G.nnz, "spmm", G.info.get("spmm"))
if os.environ.get("PROFILE"):
    from torch.profiler import profile, ProfilerActivity
    comm = FakeComm(WORLD, RANK)
    op = meld_amd.MELD(knn=15, beta=60, chebyshev_order=30, verbose=0)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        mdist.fit_transform_sharded(op, Xd, labels, comm=comm); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
if os.environ.get("CPROFILE"):  # where the HOST time of a rank's step goes (the step is issue-bound at 8 ranks)
    import cProfile, pstats
    comm = FakeComm(WORLD, RANK)
    op = meld_amd.MELD(knn=15, beta=60, chebyshev_order=30, verbose=0)
    pr = cProfile.Profile()
    pr.enable()
    mdist.fit_transform_sharded(op, Xd, labels, comm=comm); torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(30)
