"""Edge cases of the hot path against the oracle (GPU): tiny and ragged sizes, extreme
dimensions, duplicated points, badly scaled / offset data, far outliers (which blow up the
search-error allowance and push rows through the re-search and exact-sweep paths), large knn."""
import numpy as np
import pandas as pd
import pytest
import torch
from scipy import sparse

pytestmark = pytest.mark.gpu


def _check_graph(X, knn=5, decay=40, thresh=1e-4, anisotropy=1, rtol=1e-9, algorithm="brute"):
    from oracle import meld_oracle as mo

    import meld_amd

    G = mo.build_graph(np.asarray(X, dtype=np.float64), knn=knn, decay=decay, thresh=thresh, anisotropy=anisotropy, algorithm=algorithm)
    DG = meld_amd.build_knn_graph(torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).cuda(), knn=knn, decay=decay,
                                  thresh=thresh, anisotropy=anisotropy)
    A, B = sparse.csr_matrix(DG.W), sparse.csr_matrix(G.W)
    A.sort_indices(); B.sort_indices()
    assert A.nnz == B.nnz, (A.nnz, B.nnz)
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
    np.testing.assert_allclose(A.data, B.data, rtol=rtol)
    np.testing.assert_allclose(DG.dw, G.dw, rtol=rtol)
    return DG, G


@pytest.mark.parametrize("n", [3, 4, 7, 10, 63, 64, 65, 129, 255, 257, 1000])
def test_tiny_and_ragged_sizes(n):
    rng = np.random.default_rng(n)
    _check_graph(rng.normal(size=(n, 3)), knn=5)


@pytest.mark.parametrize("d", [1, 2, 15, 16, 17, 47, 48, 49, 64, 100, 126, 128])
def test_dimensions(d):
    rng = np.random.default_rng(d)
    X = rng.normal(size=(700, d))
    if d > 20:  # keep the neighbourhoods finite-sized: low intrinsic dimension embedded in d
        X = rng.normal(size=(700, 4)) @ rng.normal(size=(4, d)) + 0.01 * rng.normal(size=(700, d))
    _check_graph(X, knn=7)


def test_dimension_too_large_for_the_mfma_kernels():
    """d > 141: the split-fp16 kernel reports it loudly through the C-ABI; the graph builder then takes the
    library search path (test_wide_data_beyond_the_mfma_kernels); the fp32 kernel, asked for explicitly, fails."""
    import meld_amd
    from meld_amd._lib import get_lib
    from meld_amd.graph import HipOps

    lib = get_lib()
    assert lib.meld_knn16_kblocks(200) < 0 and b"exceeds the largest" in lib.meld_last_error()
    X = torch.from_numpy(np.random.default_rng(0).normal(size=(300, 200))).cuda()
    assert meld_amd.build_knn_graph(X).info["search"] == "wide"
    with pytest.raises(Exception, match="exceeds|unsupported|not an instantiated|padded"):
        HipOps(search="f32").directed_kernel_coo(X, 0, 300, 5, 40, 1e-4, 32)


def test_duplicated_points():
    rng = np.random.default_rng(0)
    base = rng.normal(size=(300, 4))
    X = np.concatenate([base, base[:40], base[:10]])  # 40 points twice, 10 of them three times
    DG, G = _check_graph(X, knn=5)
    assert np.isfinite(DG.dw).all()


def test_offset_and_scale_invariance_of_the_search():
    rng = np.random.default_rng(1)
    X = rng.normal(size=(1500, 10))
    # sklearn's brute-force distances use |x|^2 + |y|^2 - 2 x.y in fp64, which cancels badly under a
    # large offset; the tree searches (and the GPU refinement) use direct differences
    _check_graph(X * 1.0e4 + 3.0e6, knn=10, rtol=1e-6, algorithm="kd_tree")
    _check_graph(X * 1.0e-6 - 5.0, knn=10, rtol=1e-6, algorithm="kd_tree")


def test_far_outliers_go_through_the_fallback_paths():
    """A few points 1000x farther out than the rest inflate max|x|^2, hence the search-error
    allowance: most rows cannot be certified by the fast pass and take the re-search / exact sweep,
    and the graph must still be exact."""
    rng = np.random.default_rng(2)
    X = rng.normal(size=(2000, 6))
    X[:3] *= 1000.0
    DG, G = _check_graph(X, knn=8)
    assert DG.info["n_researched_rows"] + DG.info["n_flagged_rows"] > 0


def test_clusters_of_very_different_scale_settle_through_the_exact_bandwidth():
    """A tight cluster (spread 0.01) next to a wide one (spread 5, 30 away): inside the tight one every distance lies far
    below the error allowance of the fp16 search and of its full-split re-search, the candidate lists miss true
    neighbours, and the rows get their bandwidth from the exact recomputation (``_exact_bandwidth``) before the exact
    sweep -- which counts with its own summation order (a bandwidth rounded one ulp above the sweep's value used to flag
    11 % of such rows again and abort the build)."""
    rng = np.random.default_rng(11)
    X = np.concatenate([rng.normal(size=(2000, 20)) * 0.01, rng.normal(size=(2000, 20)) * 5.0 + 30.0])
    DG, G = _check_graph(X, knn=15, rtol=1e-9, algorithm="kd_tree")
    assert DG.info["n_flagged_rows"] >= 1000 and DG.info["n_rows_bandwidth_recomputed"] > 0


@pytest.mark.parametrize("seed", [4, 6])  # (two of the three seeds below 11 on which the library-norm routine gave up)
def test_recomputed_bandwidths_are_ranked_in_the_sweeps_own_arithmetic(seed):
    """The same situation in 52 dimensions (a fuzz case of round 4: tools/fuzz_graph.py 120 2026, case 14).  A bandwidth taken from a
    library norm differs from the sweep's own value of that distance by a few ulps at this width -- more than the two ulps the
    routine used to step down -- and the sweep then counts the bandwidth entry itself as "strictly closer", flags the row again and
    the build gave up ("could not settle the bandwidth").  The candidates' distances now come from meld_knn_pair_distances, the
    sweep's summation order: confirmed by construction."""
    rng = np.random.default_rng(seed)
    N, d = 6000, 52
    X = np.concatenate([rng.normal(size=(N // 2, d)) * 1e-3, rng.normal(size=(N - N // 2, d)) * 3 + 20])
    DG, G = _check_graph(X, knn=6, decay=10, thresh=1e-2, rtol=1e-9, algorithm="ball_tree")
    assert DG.info["n_rows_bandwidth_recomputed"] > 100


@pytest.mark.parametrize("d", [50, 52, 14, 7, 2, 100, 256, 300])
def test_the_three_exact_distance_kernels_share_one_summation_order(d):
    """refine_kernel (four lanes per candidate row for even d <= 256: slot kk & 3 takes coordinate pair kk; a lane per candidate
    otherwise), the exact sweep and meld_knn_pair_distances must produce the SAME double for the same pair of cells
    (csrc/refine.hip, head of the file): the bandwidth the refinement reports for a row is, bit for bit, the pair distance of the
    row and its knn-th neighbour, and a build whose every row goes through the sweep (force_fallback) reports the same bandwidths
    and kernel values as the build that refines candidate lists."""
    from meld_amd._lib import check, get_lib, ptr
    from meld_amd.graph import HipOps, _stream

    rng = np.random.default_rng(d)
    N, knn = 2500, 7
    X = rng.normal(size=(N, d)) * rng.uniform(0.2, 2.0, size=d) + rng.normal(size=d)
    Xd = torch.from_numpy(X).cuda()
    ops = HipOps()
    keys, vals, bw, info = ops.directed_kernel_coo(Xd, 0, N, knn, 40, 1e-4, 64)
    keys2, vals2, bw2, info2 = ops.directed_kernel_coo(Xd, 0, N, knn, 40, 1e-4, 64, force_fallback=True)
    assert info2["n_flagged_rows"] == N and (d > 64 or info["n_flagged_rows"] < N // 10)  # (in many isotropic dimensions the radius holds more cells than a candidate list)
    assert torch.equal(bw, bw2)
    a, b = ops.assemble_rows(keys, vals, 0, N, N), ops.assemble_rows(keys2, vals2, 0, N, N)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    D = torch.cdist(Xd, Xd)
    kth = torch.argsort(D, dim=1)[:, knn].contiguous()  # (self is entry 0: entry knn is the (knn + 1)-th nearest, self counted)
    rows = torch.arange(N, dtype=torch.int64, device="cuda")
    out = torch.empty(N, dtype=torch.float64, device="cuda")
    check(get_lib().meld_knn_pair_distances(ptr(Xd), d, ptr(rows), ptr(kth), N, 1, ptr(out), _stream()), "meld_knn_pair_distances")
    torch.cuda.synchronize()
    same = out == bw
    # (cdist's own rounding may order two nearly equidistant cells the other way round: a handful of rows at most)
    assert int((~same).sum()) <= 3, int((~same).sum())
    assert float(((out - bw).abs() / bw).max()) < 1e-9


@pytest.mark.parametrize("knn", [1, 2, 30, 60])
def test_knn_range(knn):
    rng = np.random.default_rng(knn)
    X = rng.normal(size=(900, 3))
    _check_graph(X, knn=knn)


def test_knn_beyond_the_candidate_list_is_rejected():
    import meld_amd

    X = torch.randn(2000, 3, dtype=torch.float64, device="cuda")
    with pytest.raises(NotImplementedError):
        meld_amd.build_knn_graph(X, knn=200)


@pytest.mark.parametrize("decay,thresh,aniso", [(10, 1e-4, 1), (40, 1e-2, 0), (2, 1e-3, 0.5), (100, 1e-6, 1)])
def test_kernel_parameters(decay, thresh, aniso):
    rng = np.random.default_rng(5)
    _check_graph(rng.normal(size=(800, 3)), knn=6, decay=decay, thresh=thresh, anisotropy=aniso, rtol=1e-8)


def test_input_containers_and_dtypes():
    import meld_amd

    rng = np.random.default_rng(3)
    X = rng.normal(size=(400, 5))
    labels = rng.choice(["a", "b"], size=400)
    ref = meld_amd.MELD(lmax=0.5).fit_transform(X, labels).values
    for alt in (X.astype(np.float32), pd.DataFrame(X), torch.from_numpy(X), torch.from_numpy(X).cuda(), np.asfortranarray(X)):
        out = meld_amd.MELD(lmax=0.5).fit_transform(alt, labels).values
        tol = 1e-4 if getattr(alt, "dtype", None) in (np.float32,) else 1e-12
        assert np.abs(out - ref).max() / np.abs(ref).max() < tol
    with pytest.raises(ValueError):
        meld_amd.MELD().fit(np.full((10, 2), np.nan))
    with pytest.raises(ValueError):
        meld_amd.MELD().fit(np.zeros(10))


def test_repeated_transform_with_many_label_sets_and_betas():
    """§8f row 3: graph and lmax stay resident; only the filter is re-run."""
    from oracle import meld_oracle as mo

    import meld_amd

    X, labels = mo.synthetic_cells(3000, n_dims=20, seed=6)
    op = meld_amd.MELD(knn=10, chebyshev_order=30).fit(X)
    G = mo.build_graph(X, knn=10, algorithm="brute")
    lmax = op.graph.lmax
    rng = np.random.default_rng(0)
    for beta in (5, 60, 150):
        lab = rng.choice(["x", "y", "z"], size=3000)
        op.set_params(beta=beta)
        out = op.transform(lab)
        samples, ind = mo.sample_indicators(lab)
        ref = mo.meld_filter(ind, G, beta=beta, chebyshev_order=30, lmax=lmax)
        assert list(out.columns) == list(samples)
        assert np.abs(out.values - ref).max() / np.abs(ref).max() < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("n,d", [(2500, 142), (1800, 300)])
def test_wide_data_beyond_the_mfma_kernels(n, d):
    """d > 141 (no PCA): the candidate search falls back to chunked library GEMMs + topk on the GPU and feeds
    the same exact refinement -- same graph and densities as the oracle."""
    import meld_amd
    from oracle import meld_oracle as mo

    X, labels = mo.synthetic_cells(n, n_dims=d, seed=17)
    op = meld_amd.MELD(knn=7, n_pca=None, chebyshev_order=30, verbose=0).fit(X)
    assert op.graph.info["search"] == "wide"
    G = mo.build_graph(X, knn=7)
    W = op.graph.W
    assert W.nnz == G.W.nnz and abs(W - G.W).max() <= 1e-9 * abs(G.W).max()
    lmax = mo.estimate_lmax(G.L, G.dw)
    op.graph.lmax = lmax
    dens = op.transform(labels)
    ref = mo.meld_filter(mo.sample_indicators(labels)[1], G, beta=60, chebyshev_order=30, lmax=lmax)
    assert np.abs(dens.values - ref).max() <= 1e-5 * np.abs(ref).max()


def test_many_uncertified_rows_trigger_one_search_with_the_longest_list():
    """A candidate list too short for the data (here forced with ksel = knn + 2; in the wild: a million cells in
    the plane) leaves most rows uncertified; instead of sweeping them one by one the builder searches once more
    with ksel = 128.  Same graph as the oracle."""
    from oracle import meld_oracle as mo

    import meld_amd

    rng = np.random.default_rng(4)
    X = rng.normal(size=(20000, 2))
    op = meld_amd.MELD(knn=10, verbose=0).fit(X, ksel=12)
    info = op.graph.info
    assert info.get("ksel_retry_from") == 12 and info["ksel"] == 128 and info["n_flagged_rows_first_try"] > 1024
    G = mo.build_graph(X, knn=10)
    W = op.graph.W
    assert W.nnz == G.W.nnz and abs(W - G.W).max() <= 1e-9 * abs(G.W).max()


def test_more_exact_duplicates_than_the_candidate_list_holds():
    """A cell with more exact copies than ksel (32 at knn = 5): its true bandwidth is 0, graphtools clips it to eps and
    the copies get K = 1 among themselves (reference path: re-search, then radius search).  The exact sweep used to
    refuse such rows ('degenerate neighbourhoods'); now they come out as the oracle builds them."""
    rng = np.random.default_rng(3)
    base = rng.normal(size=(400, 5))
    X = np.concatenate([base, np.repeat(base[:1], 45, axis=0), np.repeat(base[1:2], 70, axis=0)])
    DG, G = _check_graph(X, knn=5)
    assert np.isfinite(DG.dw).all() and DG.info["n_flagged_rows"] >= 115


def test_a_radius_that_covers_the_data_is_refused_with_an_explanation():
    """decay = 2 with thresh = 1e-6: the kernel radius is 3.7 bandwidths and in a gaussian blob covers nearly every cell -- at 150k
    cells the graph itself would be 330 GB.  The builder says so (MemoryError) instead of dying inside an allocation."""
    import meld_amd

    rng = np.random.default_rng(3)
    X = torch.from_numpy(rng.normal(size=(150_000, 12))).cuda()
    with pytest.raises(MemoryError, match="raise decay or thresh"):
        meld_amd.build_knn_graph(X, knn=18, decay=2, thresh=1e-6)
