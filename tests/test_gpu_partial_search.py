"""The search in the cells' principal frame with the partial-distance test behind the first K block (csrc/knn16.hip EE kernels,
csrc/frame.hip, HipOps.principal_frame): the graph must not depend on the frame, on the operand layout or on the test, and the
frame kernels must do what they say.  No reference counterpart ([UPSTREAM graphtools] searches the data as given): the checks are
bit-for-bit equality with the plain path of this library, which the parity tests pin on the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _frame_at_every_size(monkeypatch):
    """The product takes the principal frame from 262144 cells on (below that its fixed cost outweighs the gain); the tests want it at
    the sizes they can afford."""
    monkeypatch.setenv("MELD_KNN_ROTATE_MIN", "0")


def _cells(n, d, seed, kind="mixture"):
    rng = np.random.default_rng(seed)
    if kind == "mixture":  # a few latent dimensions embedded in d, small isotropic noise: the leading principal coordinates carry the distances
        lat = min(6, d)
        Z = rng.normal(size=(n, lat)) + rng.normal(0, 3, size=(12, lat))[rng.integers(0, 12, n)]
        Q, _ = np.linalg.qr(rng.normal(size=(d, lat)))
        return Z @ Q.T + rng.normal(0, 0.05, size=(n, d))
    if kind == "tail_heavy":  # most of the variance in FEW coordinates, but large isolated values in the others (margins of the test)
        X = rng.normal(size=(n, d)) * np.r_[np.full(4, 5.0), np.full(d - 4, 0.02)]
        X[rng.integers(0, n, 40), rng.integers(4, d, 40)] += rng.normal(0, 30.0, 40)
        return X
    if kind == "offset":  # far from the origin, tiny spread
        return 1e4 + 1e-2 * rng.normal(size=(n, d)) * np.r_[np.full(5, 50.0), np.full(d - 5, 1.0)]
    raise ValueError(kind)


def _graph(ops, Xd, knn=10):
    N = int(Xd.shape[0])
    keys, vals, bw, info = ops.directed_kernel_coo(Xd, 0, N, knn, 40, 1e-4, 64)
    return ops.assemble_rows(keys, vals, 0, N, N) + (bw,), info


@pytest.mark.parametrize("kind,d", [("mixture", 50), ("mixture", 14), ("mixture", 58), ("tail_heavy", 40), ("offset", 30), ("mixture", 20)])
def test_graph_does_not_depend_on_the_frame_or_the_partial_test(kind, d, monkeypatch):
    from meld_amd._lib import get_lib
    from meld_amd.graph import HipOps
    from meld_amd.reorder import locality_permutation

    lib = get_lib()
    assert lib.meld_knn16_split_dims(d) == 13
    X = _cells(24000, d, seed=d)
    Xd = torch.from_numpy(X).cuda()
    Xd = Xd.index_select(0, locality_permutation(Xd)).contiguous()
    ops = HipOps()
    base, info = _graph(ops, Xd)
    assert info["principal_frame"] and info["step_lists"] and info["blocks_past_partial_test"] is not None
    assert info["blocks_past_partial_test"] < 2 * info["wave_tiles_done"]  # the test dropped something
    # (a) the same frame without the test, (b) no frame (the test is then not asked for), (c) no frame with the test forced on
    monkeypatch.setenv("MELD_KNN16_EE", "0")
    a, _ = _graph(HipOps(), Xd)
    monkeypatch.delenv("MELD_KNN16_EE")
    ops_b = HipOps()
    ops_b.rotate = False
    b, info_b = _graph(ops_b, Xd)
    assert not info_b["principal_frame"] and info_b["blocks_past_partial_test"] is None
    monkeypatch.setenv("MELD_KNN16_EE", "1")
    c, _ = _graph(ops_b, Xd)
    monkeypatch.delenv("MELD_KNN16_EE")
    # (e) the two-pass form (list-filter pass + search over the thinned lists), which launches this small do not take by themselves
    monkeypatch.setenv("MELD_KNN_TWO_PHASE", "2")
    e, info_e = _graph(HipOps(), Xd)
    monkeypatch.delenv("MELD_KNN_TWO_PHASE")
    assert info_e["two_phase"] and not info["two_phase"] and 0 < info_e["pairs_past_filter"] < info_e["wave_tiles_done"]
    for other in (a, b, c, e):
        for u, v in zip(base, other):
            assert torch.equal(u, v)


def test_isotropic_cells_are_searched_as_given():
    """Full-rank isotropic data: no 13 coordinates carry half of the variance, the frame is declined and the plain pass runs."""
    from meld_amd.graph import HipOps

    X = np.random.default_rng(3).normal(size=(20000, 48))
    ops = HipOps()
    Xd = torch.from_numpy(X).cuda()
    assert ops.principal_frame(Xd, Xd.mean(0), 13) is None
    _, info = _graph(ops, Xd)
    assert not info["principal_frame"]


@pytest.mark.parametrize("n,d", [(70001, 50), (333, 7), (4096, 64)])
def test_frame_kernels(n, d):
    from meld_amd._lib import check, get_lib, ptr
    from meld_amd.graph import _stream

    lib = get_lib()
    rng = np.random.default_rng(n)
    X = rng.normal(size=(n, d)) * rng.uniform(0.1, 3.0, size=d) + rng.normal(size=d)
    Xd = torch.from_numpy(X).cuda()
    mean = Xd.mean(0)
    stride = 3
    cov = torch.zeros(d, d, dtype=torch.float64, device="cuda")
    check(lib.meld_cov_sample_f64(ptr(Xd), n, d, ptr(mean), stride, ptr(cov), _stream()), "cov")
    Xc = X[::stride] - mean.cpu().numpy()
    ref = np.triu(Xc.T @ Xc)
    assert np.abs(np.triu(cov.cpu().numpy()) - ref).max() <= 1e-10 * np.abs(ref).max()
    Qm, _ = np.linalg.qr(rng.normal(size=(d, d)))
    At = np.zeros((d, lib.meld_frame_max_dims()))
    At[:, :d] = Qm.T
    out = torch.empty_like(Xd)
    check(lib.meld_rotate_rows_f64(ptr(Xd), n, d, ptr(mean), ptr(torch.from_numpy(At).cuda()), ptr(out), _stream()), "rot")
    want = (X - mean.cpu().numpy()) @ Qm
    assert np.abs(out.cpu().numpy() - want).max() <= 1e-13 * np.abs(want).max()


def test_fit_transform_is_the_same_with_and_without_the_frame(monkeypatch):
    import meld_amd

    X = _cells(30000, 50, seed=9)
    labels = np.where(np.random.default_rng(1).random(30000) < 0.4, "a", "b")
    d1 = meld_amd.MELD(knn=7, verbose=0, lmax=2.0).fit_transform(X, labels)
    monkeypatch.setenv("MELD_KNN_ROTATE", "0")
    d0 = meld_amd.MELD(knn=7, verbose=0, lmax=2.0).fit_transform(X, labels)
    assert list(d0.columns) == list(d1.columns)
    np.testing.assert_allclose(d1.values, d0.values, rtol=0, atol=1e-12 * np.abs(d0.values).max())


@pytest.mark.parametrize("d", [32, 45])
def test_repeated_builds_agree_with_an_odd_number_of_k_blocks(d, monkeypatch):
    """Regression: with an odd number of K blocks the last staging round has planes for half of the waves only; the waves without
    one once zeroed the head of the next ring buffer under its first copy, and one build in twenty (reference slices, mid-sized
    data) lost a block's candidates.  Same cells, many builds, with and without the partial test: one graph."""
    from meld_amd.graph import HipOps
    from meld_amd.reorder import locality_permutation

    rng = np.random.default_rng(d)
    X = rng.normal(size=(33555, d)) * (10.0 ** rng.uniform(-3, 1, size=d))
    Xd = torch.from_numpy(X).cuda()
    Xd = Xd.index_select(0, locality_permutation(Xd)).contiguous()
    ops = HipOps()
    ops.rotate = False

    def bandwidths(ee):
        monkeypatch.setenv("MELD_KNN16_EE", ee)
        keys, vals, bw, info = ops.directed_kernel_coo(Xd, 0, 33555, 5, 40, 1e-2, 64)
        return bw

    ref = bandwidths("0")
    for _ in range(25):
        assert torch.equal(bandwidths("1"), ref)
        assert torch.equal(bandwidths("0"), ref)
