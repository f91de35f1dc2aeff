"""Fuzz of the partial-distance test of the search (knn16.hip EE kernels): random shapes, spectra, offsets and scales; the graph with
the test (in the principal frame, and forced on in the frame the data come in) against the graph without it, bit for bit.
python tools/fuzz_partial.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELD_DEV", "1")  # (development tool: the MELD_* switches it sets or documents are read, see meld_amd/_options.py)
import numpy as np, torch
os.environ.setdefault("MELD_KNN_ROTATE_MIN", "0")  # (the product takes the frame from 262144 cells on)
from meld_amd.graph import HipOps
from meld_amd.reorder import locality_permutation

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def graph(ops, Xd, knn, decay, thresh):
    N = int(Xd.shape[0])
    keys, vals, bw, info = ops.directed_kernel_coo(Xd, 0, N, knn, decay, thresh, 64)
    return ops.assemble_rows(keys, vals, 0, N, N) + (bw,), info


bad = 0
for c in range(n_cases):
    N = int(rng.integers(16384, 40000)); d = int(rng.integers(14, 59)); knn = int(rng.integers(2, 20))
    decay = float(rng.choice([10, 40, 100])); thresh = float(rng.choice([1e-2, 1e-4]))
    kind = rng.choice(["lowrank", "spectrum", "clusters", "spiky", "offset", "duplicates"])
    if kind == "lowrank":
        r = int(rng.integers(2, 12)); X = rng.normal(size=(N, r)) @ rng.normal(size=(r, d)) + 10.0 ** rng.uniform(-4, -1) * rng.normal(size=(N, d))
    elif kind == "spectrum":
        X = rng.normal(size=(N, d)) * (10.0 ** rng.uniform(-3, 1, size=d))
    elif kind == "clusters":
        X = rng.normal(size=(N, d)) * 0.2 + rng.normal(size=(9, d))[rng.integers(0, 9, N)] * 5 * (rng.random(d) < 0.3)
    elif kind == "spiky":
        X = rng.normal(size=(N, d)) * np.r_[np.full(5, 3.0), np.full(d - 5, 0.01)]
        X[rng.integers(0, N, 60), rng.integers(0, d, 60)] += rng.normal(0, 50.0, 60)
    elif kind == "offset":
        X = 10.0 ** rng.uniform(0, 5) + rng.normal(size=(N, d)) * (10.0 ** rng.uniform(-2, 0, size=d))
    else:
        base = rng.normal(size=(N // 4, d)) * np.r_[np.full(6, 2.0), np.full(d - 6, 0.05)]; X = base[rng.integers(0, N // 4, N)]
    if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) != c:  # (the generator has been advanced as the full run does)
        continue
    Xd = torch.from_numpy(np.ascontiguousarray(X)).cuda()
    Xd = Xd.index_select(0, locality_permutation(Xd)).contiguous()
    tag = "%-10s N=%5d d=%2d knn=%2d decay=%g thresh=%g" % (kind, N, d, knn, decay, thresh)
    for rep in range(int(os.environ.get("FUZZ_REPEAT", "1"))):
      try:
          plain = HipOps(); plain.rotate = False
          os.environ["MELD_KNN16_EE"] = "0"; ref, _ = graph(plain, Xd, knn, decay, thresh)
          os.environ["MELD_KNN16_EE"] = "1"; forced, i1 = graph(plain, Xd, knn, decay, thresh)
          del os.environ["MELD_KNN16_EE"]; framed, i2 = graph(HipOps(), Xd, knn, decay, thresh)
          # (round 6) the two-pass form -- list-filter pass + search over the thinned lists -- forced at these sizes
          os.environ["MELD_KNN_TWO_PHASE"] = "2"; two, i3 = graph(HipOps(), Xd, knn, decay, thresh); del os.environ["MELD_KNN_TWO_PHASE"]
          ok = all(torch.equal(a, b) for a, b in zip(ref, forced)) and all(torch.equal(a, b) for a, b in zip(ref, framed)) \
              and all(torch.equal(a, b) for a, b in zip(ref, two)) and (bool(i3["two_phase"]) == bool(i3["principal_frame"]))
          print("%s %s  nnz %d  past the test: forced %s, framed %s (frame %s)" % ("ok " if ok else "BAD", tag, int(ref[1].numel()),
                "%.3f" % (i1["blocks_past_partial_test"] / (2.0 * i1["wave_tiles_done"])) if i1.get("blocks_past_partial_test") is not None else "-",
                "%.3f" % (i2["blocks_past_partial_test"] / (2.0 * i2["wave_tiles_done"])) if i2.get("blocks_past_partial_test") is not None else "-",
                i2["principal_frame"]) + ("  filter kept %.3f of the pairs" % (i3["pairs_past_filter"] / max(i3["wave_tiles_done"], 1)) if i3.get("pairs_past_filter") is not None else ""), flush=True)
          bad += 0 if ok else 1
          if not ok:
              for name, other in (("forced", forced), ("framed", framed), ("two-pass", two)):
                  for what, a, b in zip(("rowptr", "col", "val", "bw"), ref, other):
                      if not torch.equal(a, b):
                          if a.shape == b.shape:
                              dif = (a != b).nonzero().flatten()
                              print("   %s: %s differs in %d places, first %s: %s vs %s" % (name, what, dif.numel(), dif[:5].tolist(), a[dif[:5]].tolist(), b[dif[:5]].tolist()))
                          else:
                              print("   %s: %s has %d entries against %d" % (name, what, b.numel(), a.numel()))
              print("   info forced:", {k: i1[k] for k in ("n_researched_rows", "n_flagged_rows") if k in i1}, " framed:", {k: i2[k] for k in ("n_researched_rows", "n_flagged_rows") if k in i2})
      except MemoryError as e:
          print("skip", tag, "(MemoryError: %s)" % str(e)[:60], flush=True)
      finally:
          os.environ.pop("MELD_KNN16_EE", None)
          os.environ.pop("MELD_KNN_TWO_PHASE", None)
print("bad:", bad)
