import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d["stages"]
print("step %.2f ms | " % d["ms_per_step"] + " ".join("%s %.2f" % (k, 1e3 * s[k]) for k in ("reorder", "prepare", "seed", "bounds", "knn_topk", "refine", "fit_total")))
