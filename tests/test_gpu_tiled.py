"""GPU tests of the panel-tiled recurrence kernel (csrc/spmm_tiled.hip): the same operator as the
CSR-stream kernel, on the re-laid-out copy of W.  Checked against the oracle's pygsp-style loop
([UPSTREAM pygsp cheby_op], reference meld/filter.py:59), against the CSR-stream kernel on the same
graph, for completeness of the layout and reproducibility, and on the shapes the row-sharded driver produces (row offset,
padded column range, empty shard)."""
import numpy as np
import pytest
import torch
from scipy import sparse

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import meld_oracle as mo

    return mo


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _graph_pair(W):
    """The same weight matrix twice: one graph forced onto the tiled layout, one on the CSR-stream kernel."""
    import meld_amd
    from meld_amd.graph import HipOps

    Gt = meld_amd.DeviceGraph.from_scipy(W)
    Gt.ops = HipOps(spmm="tiled")
    Gc = meld_amd.DeviceGraph.from_scipy(W)
    Gc.ops = HipOps(spmm="csr")
    return Gt, Gc


@pytest.fixture(scope="module")
def cells5k():
    mo = _oracle()
    X, labels = mo.synthetic_cells(5000, n_dims=50, seed=0)
    return mo.build_graph(X, knn=15, algorithm="brute")


@pytest.mark.parametrize("p", [1, 2, 3, 4, 5])
def test_tiled_recurrence_matches_oracle(cells5k, p):
    mo = _oracle()
    from meld_amd.filter import chebyshev_apply

    G = cells5k
    rng = np.random.default_rng(1)
    sig = rng.random((G.N, p))
    sig /= sig.sum(0)
    lmax = 1.01 * float(sparse.linalg.eigsh(G.L, k=1, return_eigenvectors=False)[0])
    c = mo.cheby_coeff(mo.filter_kernel_fn("heat", 60, 0, 1, lmax), lmax, 30)
    ref = mo.cheby_op(G.L, lmax, c, sig)
    Gt, Gc = _graph_pair(G.W)
    out = chebyshev_apply(Gt, torch.from_numpy(sig).cuda(), c, lmax).cpu().numpy()
    assert Gt.info["spmm"] == "tiled" and Gt.pt["nb"] >= 1
    assert _rel(out, ref) < 1e-12
    out_c = chebyshev_apply(Gc, torch.from_numpy(sig).cuda(), c, lmax).cpu().numpy()
    assert Gc.info["spmm"] == "csr"
    assert _rel(out, out_c) < 1e-13


@pytest.mark.parametrize("n,density", [(3, 1.0), (70, 0.3), (257, 0.05), (4481, 0.004), (9000, 0.002), (40000, 0.0005)])
def test_tiled_step_on_awkward_matrices(n, density):
    """Tiny graphs, a block boundary inside the row range (4481 > RMAX), empty rows, a dense row, rows longer
    than a wave; random symmetric sparsity instead of a kNN structure.  One step (y, r) and the p = 1 form
    with its dot products against a scipy evaluation, and against the CSR-stream kernel."""
    from meld_amd.graph import HipOps

    rng = np.random.default_rng(n)
    A = sparse.random(n, n, density=density, random_state=n, format="lil", dtype=np.float64)
    if n > 100:
        A[5, :] = 0  # an empty row ...
        A[:, 5] = 0
        A[7, rng.choice(n, size=min(n, 300), replace=False)] = 1.0  # ... and a long one
    A = sparse.csr_matrix(A)
    W = (A + A.T).tocsr()
    W.setdiag(0)
    W.eliminate_zeros()
    W.data = rng.random(W.nnz) + 0.1
    W = ((W + W.T) * 0.5).tocsr()
    W.sort_indices()
    if W.nnz == 0:
        W = sparse.csr_matrix(np.array([[0, 1.0, 0], [1.0, 0, 2.0], [0, 2.0, 0]]))
    Gt, Gc = _graph_pair(W)
    dw = np.ravel(W.sum(1))
    for p in (1, 2, 3):
        x = rng.normal(size=(n, p))
        z = rng.normal(size=(n, p))
        r0 = rng.normal(size=(n, p))
        al, be, ga, co = 0.7, -0.3, -1.0, 0.25
        y_ref = al * (dw[:, None] * x - W @ x) + be * x + ga * z
        r_ref = r0 + co * y_ref
        outs = []
        for G in (Gt, Gc):
            xd, zd = torch.from_numpy(x).cuda(), torch.from_numpy(z).cuda()
            yd, rd = torch.empty_like(xd), torch.from_numpy(r0.copy()).cuda()
            dots = torch.zeros(2 * G.ops.dot_slots(), dtype=torch.float64, device="cuda") if p == 1 else None
            G.ops.cheby_step(G, p, xd, 0, zd, yd, rd, al, be, ga, co, dots)
            torch.cuda.synchronize()
            assert _rel(yd.cpu().numpy(), y_ref) < 1e-13
            assert _rel(rd.cpu().numpy(), r_ref) < 1e-13
            if p == 1:
                s = G.ops.dot_slots()
                d = dots.cpu().numpy()
                assert abs(d[:s].sum() - float((y_ref * x).sum())) <= 1e-11 * max(1.0, abs(float((y_ref * x).sum())))
                assert abs(d[s:].sum() - float((y_ref * y_ref).sum())) <= 1e-11 * float((y_ref * y_ref).sum())
            outs.append(yd.cpu().numpy())
        assert Gt.info["spmm"] == "tiled"
        assert _rel(outs[0], outs[1]) < 1e-13


def test_tiled_layout_holds_every_nonzero_once_and_is_reproducible():
    """Every nonzero of W appears exactly once in the layout -- OUT entries decoded back through the column lists, IN
    pairs (stored once per symmetric pair) mirrored -- and two independent builds + runs agree to rounding (several
    waves add into one LDS accumulator, so the order of the additions is not fixed: not bit for bit)."""
    import meld_amd
    from meld_amd.graph import HipOps
    from meld_amd._lib import get_lib
    import ctypes as C

    mo = _oracle()
    X, _ = mo.synthetic_cells(20000, n_dims=50, seed=5)
    G = meld_amd.build_knn_graph(torch.from_numpy(X).cuda(), knn=15)
    G.ops = HipOps(spmm="tiled")
    pt = G.ops.pt_layout(G)
    assert pt is not None and G.info["spmm_fold"] is True
    t = {k: v.cpu().numpy() for k, v in pt["tensors"].items()}
    nw, rmax, cp, tmax = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    get_lib().meld_pt_geometry(C.byref(nw), C.byref(rmax), C.byref(cp), C.byref(tmax))
    nw, cp, rmax = nw.value, cp.value, rmax.value
    slots, segw, segrows, padcap = rmax // nw, 64, 2 * (nw + 1), 64
    rowptr = G.rowptr.cpu().numpy()
    nb = pt["nb"]
    blk_row = t["blk_row"]
    assert blk_row[0] == 0 and blk_row[nb] == G.n_rows and np.all(np.diff(blk_row) >= 0) and np.diff(blk_row).max() <= rmax
    pidx = t["pidx"].view(np.uint32)

    def row_of(slot):  # inverse of the kernel's row_slot()
        return (slot % slots) * nw + slot // slots

    rows, cols, vals = [], [], []
    n_pairs = 0
    for b in range(nb):
        segp = t["seg"][b * segrows * segw : (b + 1) * segrows * segw].reshape(segrows, segw)
        T, e0, r0 = int(t["blk_ntile"][b]), int(rowptr[blk_row[b]]), int(blk_row[b])
        assert segp[segrows - 1, 2] == r0 and segp[segrows - 1, 4] == T
        lst = t["list_cols"][e0 : e0 + int(t["blk_ndist"][b])]
        assert np.all(np.diff(lst) > 0)  # sorted distinct columns
        base_b = e0 + b * nw * padcap
        for w in range(nw):
            hdr = int(np.uint32(segp[w, 62]))
            n_in, soff = hdr & 0xFFFF, int(segp[w, 63])
            eoff = segp[nw + 1 + w]
            sb = base_b + soff
            # IN pairs: whole chunks, padded with (0, 0) entries
            ix, v = pidx[sb : sb + 64 * n_in], t["pval"][sb : sb + 64 * n_in]
            keep = v != 0
            ri, rj = row_of((ix[keep] >> 20).astype(np.int64)), row_of(((ix[keep] >> 4) & 0xFFF).astype(np.int64))
            assert keep.sum() == eoff[62] and np.all(ri < rj) and np.all(ri % nw == w)
            rows += [r0 + ri, r0 + rj]
            cols += [r0 + rj, r0 + ri]
            vals += [v[keep], v[keep]]
            n_pairs += int(keep.sum())
            # OUT entries: dense, tile by tile
            for j in range(T):
                s, e = sb + 64 * n_in + int(eoff[j]), sb + 64 * n_in + int(eoff[j + 1] if j + 1 < T else eoff[63])
                ix = pidx[s:e]
                cs = ((ix >> 4) & 0xFFF).astype(np.int64)
                assert np.all(cs >> 10 == (j & 3))
                rl = row_of((ix >> 20).astype(np.int64))
                assert np.all(rl % nw == w)
                rows.append(r0 + rl)
                cols.append(lst[int(segp[nw, j]) * cp + (cs & (cp - 1))])  # row nw of seg: list chunk of the j-th processed tile
                vals.append(t["pval"][s:e])
    M = sparse.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(G.n_rows, G.N))
    ref = sparse.csr_matrix((G.val.cpu().numpy(), G.col.cpu().numpy(), rowptr), shape=(G.n_rows, G.N))
    assert M.nnz == ref.nnz and abs(M - ref).max() == 0.0
    assert n_pairs > 0.1 * ref.nnz  # the fold is doing something on a kNN graph in locality order
    # reproducible to rounding
    x = torch.rand(G.N, 2, dtype=torch.float64, device="cuda")
    ys = []
    for _ in range(2):
        G.pt = None
        y = torch.empty_like(x)
        G.ops.cheby_step(G, 2, x, 0, None, y, None, 1.0, 0.0, 0.0, 0.0)
        ys.append(y.cpu().numpy().copy())
    assert _rel(ys[0], ys[1]) < 1e-14


def test_a_weight_matrix_that_is_not_symmetric_is_not_folded():
    """The fold stores one value per in-block pair: a W whose two halves differ (uploaded from elsewhere) must be
    detected by the builder's symmetry check and laid out unfolded -- same results as the CSR-stream kernel."""
    rng = np.random.default_rng(3)
    n = 3000
    A = sparse.random(n, n, density=0.01, random_state=3, format="csr", dtype=np.float64)
    A.setdiag(0)
    A.eliminate_zeros()
    A.data = rng.random(A.nnz) + 0.1
    S = (A + A.T).tocsr()  # symmetric pattern ...
    S.data = rng.random(S.nnz) + 0.1  # ... with unrelated values in the two halves
    S.sort_indices()
    Gt, Gc = _graph_pair(S)
    x = rng.normal(size=(n, 2))
    outs = []
    for G in (Gt, Gc):
        y = torch.empty(n, 2, dtype=torch.float64, device="cuda")
        G.ops.cheby_step(G, 2, torch.from_numpy(x).cuda(), 0, None, y, None, 1.0, 0.0, 0.0, 0.0)
        outs.append(y.cpu().numpy())
    assert Gt.info["spmm"] == "tiled" and Gt.info["spmm_fold"] is False
    ref = np.ravel(S.sum(1))[:, None] * x - S @ x
    assert _rel(outs[0], ref) < 1e-13 and _rel(outs[1], ref) < 1e-13
    # the symmetric matrix with the same pattern is folded
    Gs, _ = _graph_pair(((S + S.T) * 0.5).tocsr())
    y = torch.empty(n, 2, dtype=torch.float64, device="cuda")
    Gs.ops.cheby_step(Gs, 2, torch.from_numpy(x).cuda(), 0, None, y, None, 1.0, 0.0, 0.0, 0.0)
    assert Gs.info["spmm_fold"] is True


def test_tiled_kernel_on_a_row_shard():
    """What a rank of the row-sharded driver holds: local rows [r0, r1) of W with global column indices into a
    padded iterate, x_row_offset = r0; and a shard without rows."""
    import meld_amd
    from meld_amd.graph import DeviceGraph, HipOps

    mo = _oracle()
    X, _ = mo.synthetic_cells(6000, n_dims=20, seed=9)
    W = mo.build_graph(X, knn=10, algorithm="brute").W.tocsr()
    n, r0, r1, n_pad = 6000, 2048, 4096, 6144
    Wl = W[r0:r1]
    dev = "cuda"
    G = DeviceGraph(torch.from_numpy(Wl.indptr.astype(np.int64)).to(dev), torch.from_numpy(Wl.indices.astype(np.int32)).to(dev),
                    torch.from_numpy(Wl.data).to(dev), torch.from_numpy(np.ravel(Wl.sum(1))).to(dev), row_begin=r0, n_total=n)
    G.rows_pad, G.n_pad = r1 - r0, n_pad
    G.ops = HipOps(spmm="tiled")
    rng = np.random.default_rng(0)
    xf = np.zeros((n_pad, 2))
    xf[:n] = rng.normal(size=(n, 2))
    xd = torch.from_numpy(xf).cuda()
    y = torch.empty(r1 - r0, 2, dtype=torch.float64, device=dev)
    G.ops.cheby_step(G, 2, xd, r0, None, y, None, 2.0, 0.5, 0.0, 0.0)
    ref = 2.0 * (np.ravel(Wl.sum(1))[:, None] * xf[r0:r1] - Wl @ xf[:n]) + 0.5 * xf[r0:r1]
    assert G.info["spmm"] == "tiled"
    assert _rel(y.cpu().numpy(), ref) < 1e-13
    # the Lanczos SpMV phase on the shard (p = 1, scalars from device memory, dots accumulated)
    state = torch.zeros(8, dtype=torch.float64, device=dev)
    state[3], state[4] = 0.5, -0.25
    dots = torch.zeros(2 * G.ops.dot_slots(), dtype=torch.float64, device=dev)
    x1 = xd[:, 0].contiguous()
    z1 = torch.from_numpy(rng.normal(size=r1 - r0)).cuda()
    y1 = torch.empty(r1 - r0, dtype=torch.float64, device=dev)
    G.ops.lanczos_spmv(G, x1, z1, y1, state, dots)
    ref1 = 0.5 * (np.ravel(Wl.sum(1)) * xf[r0:r1, 0] - Wl @ xf[:n, 0]) - 0.25 * z1.cpu().numpy()
    assert _rel(y1.cpu().numpy(), ref1) < 1e-6  # (the lmax estimate's SpMV streams the fp32 copy of the weights)
    s = G.ops.dot_slots()
    assert abs(dots[:s].sum().item() - float(ref1 @ xf[r0:r1, 0])) < 1e-5 * abs(float(ref1 @ xf[r0:r1, 0])) + 1e-9
    # an empty shard: nothing launched, nothing raised
    E = DeviceGraph(torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int32, device=dev),
                    torch.zeros(0, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev),
                    row_begin=n, n_total=n)
    E.n_rows = 0
    E.ops = HipOps(spmm="tiled")
    E.ops.cheby_step(E, 2, xd, n, None, y, None, 1.0, 0.0, 0.0, 0.0)
    E.ops.lanczos_spmv(E, x1, z1, y1, state, dots)
    torch.cuda.synchronize()


def test_lmax_and_densities_do_not_depend_on_the_recurrence_kernel():
    """End to end at a size where the default picks the tiled layout: lmax and the densities of the two kernels agree
    to rounding, and the Lanczos estimate is the converged eigenvalue."""
    import meld_amd
    from meld_amd.graph import HipOps
    from bench import synthetic_cells

    X, labels = synthetic_cells(80000, 50, seed=2)
    op = meld_amd.MELD(knn=15, chebyshev_order=30)
    out = op.fit_transform(X, labels)
    G = op.graph
    assert G.info["spmm"] == "tiled"
    lm_t = G.lmax
    G2 = meld_amd.DeviceGraph(G.rowptr, G.col, G.val, G.dw_dev, ksum=G.ksum, anisotropy=G.anisotropy)
    G2.perm = G.perm
    G2.ops = HipOps(spmm="csr")
    op2 = meld_amd.MELD(knn=15, chebyshev_order=30).fit(G2)
    out2 = op2.transform(labels)
    assert G2.info["spmm"] == "csr"
    # the tiled Lanczos streams the fp32 copy of the weights: the two estimates agree to ~1e-7, both are within the
    # Lanczos tolerance's eigenvalue error of the converged value
    assert abs(G2.lmax - lm_t) <= 1e-6 * lm_t
    lam = float(sparse.linalg.eigsh(G.L, k=1, tol=1e-10, return_eigenvectors=False)[0])
    assert abs(lm_t / 1.01 - lam) <= 2e-5 * lam and abs(G2.lmax / 1.01 - lam) <= 2e-5 * lam
    # ... and with the same lmax the two recurrence kernels give the same densities to rounding
    G2.lmax = lm_t
    out2 = op2.transform(labels)
    assert np.abs(out.values - out2.values).max() <= 1e-12 * np.abs(out2.values).max()


def test_graphs_the_layout_refuses_stay_on_the_csr_kernel():
    """A block of a random (non-kNN) graph can touch more distinct columns than 63 tiles hold: the builder reports it
    (status 2), the graph stays on the CSR-stream kernel, results unchanged."""
    from meld_amd.graph import HipOps
    import meld_amd

    n = 200_000
    rng = np.random.default_rng(0)
    rows = np.repeat(np.arange(n), 64)
    cols = rng.integers(0, n, size=rows.shape[0])
    A = sparse.csr_matrix((rng.random(rows.shape[0]) + 0.1, (rows, cols)), shape=(n, n))
    W = ((A + A.T) * 0.5).tocsr()
    W.setdiag(0)
    W.eliminate_zeros()
    W.sort_indices()
    G = meld_amd.DeviceGraph.from_scipy(W)
    G.ops = HipOps(spmm="tiled")
    x = rng.normal(size=(n, 2))
    y = torch.empty(n, 2, dtype=torch.float64, device="cuda")
    G.ops.cheby_step(G, 2, torch.from_numpy(x).cuda(), 0, None, y, None, 1.0, 0.0, 0.0, 0.0)
    assert G.info["spmm"].startswith("csr (tiled layout refused")
    ref = np.ravel(W.sum(1))[:, None] * x - W @ x
    assert _rel(y.cpu().numpy(), ref) < 1e-13

