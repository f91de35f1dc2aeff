"""(K + K^T)/2 assembly: the row-bucket path (one wave sorts and merges a row; include/meld_hip.h,
meld_coo_row_counts ... meld_csr_compact_rows) against the global sort + reduce-by-key path and against scipy's
coo -> csr (duplicates summed), which is what graphtools' symmetrize_kernel does on scipy.sparse
([UPSTREAM graphtools BaseGraph.symmetrize_kernel], called under reference meld/meld.py:117-118)."""
import numpy as np
import pytest
import torch
from scipy import sparse

pytestmark = pytest.mark.gpu


def _assemble(keys, vals, row_begin, n_rows, N, mode, monkeypatch):
    from meld_amd.graph import HipOps

    monkeypatch.setenv("MELD_ASSEMBLE", mode)
    ops = HipOps()
    rp, col, val = ops.assemble_rows(torch.from_numpy(keys).cuda(), torch.from_numpy(vals).cuda(), row_begin, n_rows, N)
    torch.cuda.synchronize()
    return rp.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()


def _sym_coo(rng, N, deg, hub=0):
    """directed entries (i, j, v), i != j, unique per (i, j); emitted in both directions at v / 2 like meld_coo_emit"""
    i = np.repeat(np.arange(N), deg)
    j = rng.integers(0, N, size=i.shape[0])
    if hub:
        j[: hub] = 0  # many rows point at cell 0: a long transposed row
    keep = i != j
    i, j = i[keep], j[keep]
    _, first = np.unique(i.astype(np.int64) << 32 | j, return_index=True)
    i, j = i[first], j[first]
    v = rng.random(i.shape[0])
    keys = np.concatenate([(i.astype(np.int64) << 32) | j, (j.astype(np.int64) << 32) | i])
    vals = np.concatenate([0.5 * v, 0.5 * v])
    p = rng.permutation(keys.shape[0])
    return keys[p], vals[p]


@pytest.mark.parametrize("N,deg", [(1000, 7), (20000, 20), (5000, 70)])
def test_bucket_assembly_equals_sort_assembly_and_scipy(N, deg, monkeypatch):
    rng = np.random.default_rng(N)
    keys, vals = _sym_coo(rng, N, deg)
    rb, cb, vb = _assemble(keys, vals, 0, N, N, "bucket", monkeypatch)
    rs, cs, vs = _assemble(keys, vals, 0, N, N, "sort", monkeypatch)
    assert np.array_equal(rb, rs) and np.array_equal(cb, cs)
    assert np.array_equal(vb, vs)  # bit for bit: at most two addends per entry
    W = sparse.coo_matrix((vals, (keys >> 32, keys & 0xFFFFFFFF)), shape=(N, N)).tocsr()
    W.sum_duplicates()
    W.sort_indices()
    assert np.array_equal(rb, W.indptr) and np.array_equal(cb, W.indices)
    np.testing.assert_allclose(vb, W.data, rtol=1e-15)


def test_bucket_assembly_of_a_row_slice(monkeypatch):
    """the sharded driver assembles rows [row_begin, row_begin + n_rows) only; other keys are ignored"""
    rng = np.random.default_rng(3)
    N = 6000
    keys, vals = _sym_coo(rng, N, 12)
    for mode in ("bucket", "sort"):
        sel = (keys >> 32 >= 1500) & (keys >> 32 < 4000)
        r, c, v = _assemble(keys[sel], vals[sel], 1500, 2500, N, mode, monkeypatch)
        W = sparse.coo_matrix((vals, (keys >> 32, keys & 0xFFFFFFFF)), shape=(N, N)).tocsr()[1500:4000]
        W.sum_duplicates()
        W.sort_indices()
        assert np.array_equal(r, W.indptr) and np.array_equal(c, W.indices)
        np.testing.assert_allclose(v, W.data, rtol=1e-15)
    # keys outside the slice are skipped by the bucket path
    r2, c2, v2 = _assemble(keys, vals, 1500, 2500, N, "bucket", monkeypatch)
    assert np.array_equal(r2, r) and np.array_equal(c2, c) and np.array_equal(v2, v)


def test_columns_beyond_24_bits_take_the_wide_keys(monkeypatch):
    """One wave sorts a row on 32-bit (column : slot) keys when every column of the row is below 2^24 - 1 and on 64-bit keys
    otherwise -- decided per row.  A row slice of a graph of 20 M cells: some rows hold small columns only, some large ones only,
    most a mix; both paths against scipy and against the sort-based assembly."""
    rng = np.random.default_rng(24)
    N, row_begin, n_rows = 20_000_000, 1000, 3000
    i = np.repeat(np.arange(row_begin, row_begin + n_rows), 30)
    kind = rng.integers(0, 3, size=n_rows)[i - row_begin]  # 0: small columns, 1: beyond 2^24, 2: both
    small = rng.integers(0, (1 << 24) - 1, size=i.shape[0])
    large = rng.integers((1 << 24) - 1, N, size=i.shape[0])
    j = np.where(kind == 0, small, np.where(kind == 1, large, np.where(rng.random(i.shape[0]) < 0.5, small, large)))
    j[::7] = (1 << 24) - 1 + (j[::7] % 3) - 1  # the boundary itself: 2^24 - 2 (narrow), 2^24 - 1 and 2^24 (wide)
    keys = (i.astype(np.int64) << 32) | j
    keys, first = np.unique(keys, return_index=True)
    vals = rng.random(keys.shape[0])
    dup = rng.random(keys.shape[0]) < 0.3  # a second entry of the same (row, column), as the transposed copy of a mutual pair is
    keys = np.concatenate([keys, keys[dup]])
    vals = np.concatenate([vals, rng.random(int(dup.sum()))])
    p = rng.permutation(keys.shape[0])
    keys, vals = keys[p], vals[p]
    rb, cb, vb = _assemble(keys, vals, row_begin, n_rows, N, "bucket", monkeypatch)
    rs, cs, vs = _assemble(keys, vals, row_begin, n_rows, N, "sort", monkeypatch)
    assert np.array_equal(rb, rs) and np.array_equal(cb, cs) and np.array_equal(vb, vs)
    W = sparse.coo_matrix((vals, ((keys >> 32) - row_begin, keys & 0xFFFFFFFF)), shape=(n_rows, N)).tocsr()
    W.sum_duplicates()
    W.sort_indices()
    assert np.array_equal(rb, W.indptr) and np.array_equal(cb, W.indices)
    np.testing.assert_allclose(vb, W.data, rtol=1e-15)


def test_long_rows_and_repeated_keys_take_the_sort_path(monkeypatch):
    """a hub row of more than 256 entries, and a key that occurs three times: the bucket path must hand over to the
    sort-based one (same result as asking for it)"""
    rng = np.random.default_rng(5)
    N = 4000
    keys, vals = _sym_coo(rng, N, 6, hub=3000)
    assert np.bincount((keys >> 32).astype(np.int64)).max() > 256
    rb, cb, vb = _assemble(keys, vals, 0, N, N, "bucket", monkeypatch)
    rs, cs, vs = _assemble(keys, vals, 0, N, N, "sort", monkeypatch)
    assert np.array_equal(rb, rs) and np.array_equal(cb, cs) and np.array_equal(vb, vs)
    keys3, vals3 = _sym_coo(rng, N, 6)
    keys3 = np.concatenate([keys3, keys3[:50]])
    vals3 = np.concatenate([vals3, rng.random(50)])
    rb, cb, vb = _assemble(keys3, vals3, 0, N, N, "bucket", monkeypatch)
    rs, cs, vs = _assemble(keys3, vals3, 0, N, N, "sort", monkeypatch)
    assert np.array_equal(rb, rs) and np.array_equal(cb, cs) and np.array_equal(vb, vs)


@pytest.mark.parametrize("n,force", [(30000, False), (6000, True)])
def test_candidates_straight_into_the_buckets_build_the_same_graph(n, force, monkeypatch):
    """meld_coo_emit_scatter (single GPU: the kept candidates go into the row buckets without the COO detour) builds bit for
    bit the CSR of the emit + scatter path -- also when every row comes from the exact sweep (force_fallback)."""
    import meld_amd

    rng = np.random.default_rng(5)
    X = torch.from_numpy(rng.normal(size=(n, 12))).cuda()
    monkeypatch.setenv("MELD_ASSEMBLE_FUSED", "1")
    A = meld_amd.build_knn_graph(X, knn=9, force_fallback=force)
    monkeypatch.setenv("MELD_ASSEMBLE_FUSED", "0")
    B = meld_amd.build_knn_graph(X, knn=9, force_fallback=force)
    assert torch.equal(A.rowptr, B.rowptr) and torch.equal(A.col, B.col) and torch.equal(A.val, B.val)
    assert torch.equal(A.dw_dev, B.dw_dev)


@pytest.mark.parametrize("a", [1.0, 0.5, 0.0])
def test_anisotropy_and_degrees_in_one_pass_equal_the_two_kernels(a):
    """meld_csr_anisotropy_degrees = meld_csr_anisotropy followed by meld_csr_row_sums(diag = 0), bit for bit (same lanes, same
    order of additions), on rows of every length from empty to a few hundred."""
    from meld_amd.graph import HipOps

    rng = np.random.default_rng(5)
    n = 5000
    lens = rng.integers(0, 60, size=n)
    lens[::97] = 0
    lens[5::211] = rng.integers(100, 400, size=lens[5::211].shape[0])
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    col = rng.integers(0, n, size=int(rp[-1])).astype(np.int32)
    val = rng.random(int(rp[-1]))
    ops = HipOps()
    rowptr, cols = torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda()
    v1, v2 = torch.from_numpy(val).cuda(), torch.from_numpy(val.copy()).cuda()
    ksum = ops.row_sums(rowptr, v1, n, 1.0)
    ops.anisotropy(rowptr, cols, v1, n, ksum, 0, a)
    dw1 = ops.row_sums(rowptr, v1, n, 0.0)
    dw2 = ops.anisotropy_degrees(rowptr, cols, v2, n, ksum, 0, a)
    torch.cuda.synchronize()
    assert torch.equal(v1, v2) and torch.equal(dw1, dw2)
    ks = ksum.cpu().numpy()
    rows = np.repeat(np.arange(n), lens)
    np.testing.assert_allclose(v2.cpu().numpy(), val / (ks[rows] * ks[col]) ** a, rtol=1e-14)


def test_row_sums_out_of_the_compaction_equal_the_row_sum_kernel():
    """meld_csr_compact_rows_sums = meld_csr_compact_rows followed by meld_csr_row_sums, bit for bit (eight lanes per row in the
    row-sum kernel's order), empty rows and rows of a few hundred entries included."""
    from meld_amd._lib import check, get_lib, ptr
    from meld_amd.graph import HipOps

    lib = get_lib()
    B = int(lib.meld_csr_bucket_slots())
    rng = np.random.default_rng(8)
    n = 3000
    lens = rng.integers(0, 70, size=n)
    lens[::53] = 0
    lens[7::301] = B
    rp = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).cuda()
    tcol = torch.from_numpy(rng.integers(0, n, size=n * B).astype(np.int32)).cuda()
    tval = torch.from_numpy(rng.random(n * B)).cuda()
    nnz = int(rp[-1])
    st = torch.cuda.current_stream().cuda_stream
    c1, v1 = torch.empty(nnz, dtype=torch.int32, device="cuda"), torch.empty(nnz, dtype=torch.float64, device="cuda")
    c2, v2, s2 = torch.empty_like(c1), torch.empty_like(v1), torch.empty(n, dtype=torch.float64, device="cuda")
    check(lib.meld_csr_compact_rows(ptr(rp), n, ptr(tcol), ptr(tval), ptr(c1), ptr(v1), st))
    check(lib.meld_csr_compact_rows_sums(ptr(rp), n, ptr(tcol), ptr(tval), ptr(c2), ptr(v2), 1.0, ptr(s2), st))
    s1 = HipOps().row_sums(rp, v1, n, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(c1, c2) and torch.equal(v1, v2) and torch.equal(s1, s2)


@pytest.mark.parametrize("mode,theta", [(1, 0.0), (2, 1.0), (2, 0.35), (2, 0.0)])
def test_symmetrisation_modes_of_the_bucket_merge(mode, theta):
    """meld_csr_rows_sort_merge with symm = 1 ("*": K o K^T) and 2 ("mnn": theta min + (1 - theta) max, a missing direction
    counting as 0) against the same formulas on scipy matrices; the result stays bitwise symmetric."""
    from meld_amd.graph import HipOps

    rng = np.random.default_rng(7)
    N = 9000
    i = np.repeat(np.arange(N), 15)
    j = rng.integers(0, N, size=i.shape[0])
    j[: N * 5] = (i[: N * 5] + rng.integers(1, 6, size=N * 5)) % N  # plenty of mutual pairs
    keep = i != j
    i, j = i[keep], j[keep]
    _, first = np.unique(i.astype(np.int64) << 32 | j, return_index=True)
    i, j = i[first], j[first]
    v = rng.random(i.shape[0]) + 0.01
    K = sparse.coo_matrix((v, (i, j)), shape=(N, N)).tocsr()
    keys = np.concatenate([(i.astype(np.int64) << 32) | j, (j.astype(np.int64) << 32) | i])
    vals = np.concatenate([0.5 * v, 0.5 * v])
    rp, col, val = HipOps().assemble_rows(torch.from_numpy(keys).cuda(), torch.from_numpy(vals).cuda(), 0, N, N, symm=(mode, theta))
    got = sparse.csr_matrix((val.cpu().numpy(), col.cpu().numpy(), rp.cpu().numpy()), shape=(N, N))
    if mode == 1:
        ref = K.multiply(K.T).tocsr()
    else:
        ref = (theta * K.minimum(K.T) + (1 - theta) * K.maximum(K.T)).tocsr()
    ref.eliminate_zeros()
    ref.sort_indices()
    assert np.array_equal(got.indptr, ref.indptr) and np.array_equal(got.indices, ref.indices)
    np.testing.assert_allclose(got.data, ref.data, rtol=1e-14)
    assert abs(got - got.T).max() == 0
