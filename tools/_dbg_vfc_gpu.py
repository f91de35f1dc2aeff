import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meld_amd
rng = np.random.default_rng(9)
X = rng.normal(size=(600, 5)) * np.array([3.0, 2.0, 1.5, 1.0, 0.5])
labels = np.where(X[:, 0] + 0.7 * rng.normal(size=600) > 0, "expt", "ctrl")
op = meld_amd.MELD(knn=7, verbose=0); op.fit_transform(X, labels)
G = op.graph
for R in (64, 1024):
    vfc = meld_amd.VertexFrequencyCluster(method="filterbank", n_probes=R, n_bands=12, random_state=3).fit(G)
    est = vfc._fb_spectrogram.cpu().numpy()
    L = np.asarray(G.L.todense()); lam, U = np.linalg.eigh(L); lam = np.clip(lam, 0, None)
    lmax = vfc._fb["lmax"]; T, B = len(vfc.window_sizes), vfc.n_bands
    P = np.polynomial.chebyshev.chebval(2.0 * lam / lmax - 1.0, vfc._fb["coeffs"].T)
    E = np.clip((U * U) @ P.T, 0, None).reshape(-1, T, B)
    ref = np.tanh(np.sqrt(E / E.sum(2, keepdims=True))).sum(1)
    print("R", R, "order", vfc._fb["order"], "mean err", np.abs(est - ref).mean(), "max", np.abs(est - ref).max(), "mean ref", ref.mean())
    th = vfc._fb["ritz"].cpu().numpy()
    print("  ritz[:5]", th[:5], "exact[:5]", lam[:5], "lmax", lmax, "finite", np.isfinite(est).all())
    print("  est0", est[0][:4], "ref0", ref[0][:4])
