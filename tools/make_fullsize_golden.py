"""Oracle digests at BASELINE.json's full sizes: the WHOLE CPU oracle (oracle/meld_oracle.py: brute-force kNN on all host cores,
alpha-decay kernel, symmetrisation, anisotropy, Chebyshev filter -- [UPSTREAM graphtools ``build_kernel_to_data`` / pygsp
``cheby_op``] as reached from reference meld/meld.py:273 and meld/filter.py:59) run once at C3 (500k x 50) and at the 1M x 50 size
of C4 on the GPU box's host cores (50 s / 170 s on 256 cores; it is the checker there), reduced to a fixture small enough to
commit: ``tests/golden/g8_fullsize.npz``.  ``tests/test_gpu_fullsize.py::test_whole_path_against_the_oracle_digest`` compares
the HIP build with it in seconds, in the driver-run tier.

    gpurun -- 'python tools/make_fullsize_golden.py gpurun_out/g8_fullsize.npz [500000,1000000]'   (then copy to tests/golden/)

Per size N (keys prefixed ``n<N>_``), everything in the INPUT order of the cells, CSR in canonical (sorted-column) form:
  nnz, lmax (the oracle's own estimate: the tests inject it), sha256 of W.indptr (int64) and of W.indices (int32) -- the sparsity
  pattern is compared exactly; for the floating-point arrays W.data, dw, bandwidth and the two density columns: K seeded +-1
  projections (fp64 dot products: any entry that moves by e moves every projection by e), the l2 norm, and the values at 4096
  seeded sample positions.  No reference source, no oracle output beyond these numbers."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np
from scipy import sparse

N_PROJ, N_SAMPLE = 8, 4096


def digest_vector(v, seed):
    """(projections[N_PROJ], l2 norm, sample positions, sample values) of a 1-D fp64 array; the same ``seed`` and length give the
    same signs and positions on any machine (numpy's PCG64 streams are platform-independent)."""
    v = np.ascontiguousarray(v, dtype=np.float64).ravel()
    rng = np.random.default_rng(seed)
    proj = np.empty(N_PROJ)
    for k in range(N_PROJ):
        s = rng.integers(0, 2, size=v.shape[0], dtype=np.int8)
        proj[k] = float(np.dot(v, s.astype(np.float64) * 2.0 - 1.0))
    pos = np.sort(rng.choice(v.shape[0], size=min(N_SAMPLE, v.shape[0]), replace=False))
    return proj, float(np.sqrt(np.dot(v, v))), pos.astype(np.int64), v[pos].copy()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def graph_digest(prefix, W, dw, bandwidth, dens, lmax, out):
    """The fixture entries of one size (shared with the test, which calls it on the device-built graph)."""
    W = sparse.csr_matrix(W)
    W.sort_indices()
    out[prefix + "nnz"] = np.int64(W.nnz)
    out[prefix + "lmax"] = np.float64(lmax)
    out[prefix + "sha_indptr"] = np.array(sha(W.indptr.astype(np.int64)))
    out[prefix + "sha_indices"] = np.array(sha(W.indices.astype(np.int32)))
    vecs = {"wdata": W.data, "dw": np.ravel(dw), "bandwidth": np.ravel(bandwidth)}
    for c in range(dens.shape[1]):
        vecs["dens%d" % c] = dens[:, c]
    for i, (name, v) in enumerate(sorted(vecs.items())):
        proj, nrm, pos, val = digest_vector(v, 1000 + i)
        out[prefix + name + "_proj"] = proj
        out[prefix + name + "_norm"] = np.float64(nrm)
        out[prefix + name + "_pos"] = pos
        out[prefix + name + "_val"] = val
    return out


def main():
    from oracle import meld_oracle as mo

    path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/g8_fullsize.npz"
    sizes = [int(s) for s in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["500000", "1000000"])]
    out = {"sizes": np.array(sizes, dtype=np.int64), "cores": np.int64(os.cpu_count()), "commit": np.array(os.environ.get("MELD_COMMIT", "unknown"))}
    for n in sizes:
        X, labels = mo.synthetic_cells(n, n_dims=50, seed=0)
        t0 = time.perf_counter()
        G = mo.build_graph(X, knn=15, decay=40, thresh=1e-4, anisotropy=1, algorithm="brute", n_jobs=-1)
        samples, ind = mo.sample_indicators(labels)
        dens = mo.meld_filter(ind, G, beta=60, chebyshev_order=30)
        secs = time.perf_counter() - t0
        graph_digest("n%d_" % n, G.W, G.dw, G.info["bandwidth"], dens, G.lmax, out)
        out["n%d_samples" % n] = np.array([str(s) for s in samples])
        out["n%d_oracle_seconds" % n] = np.float64(secs)
        print("N=%d: oracle %.1f s on %d cores, nnz %d, lmax %.12g" % (n, secs, os.cpu_count(), int(out["n%d_nnz" % n]), G.lmax), flush=True)
        del X, G, dens
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
