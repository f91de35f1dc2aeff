#!/usr/bin/env python
"""HBM-side traffic per launch of the hot kernels from a rocprofv3 --pmc pass (run on the GPU box):

    python tools/pmc_traffic.py [--cells 1000000] [--out gpurun_out/traffic.json]

One counters-only pass (`--kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum`, no other tracing domain) over
one bench step; bytes = RDREQ x 128 B + WRREQ x 64 B.  (MI355X_MICROARCH.md: the fabric-side read counter
tallies 128-byte requests at 64 B on gfx950 -- FETCH_SIZE = RDREQ x 64 B is half the bytes of a wide coalesced
read; calibrated here on the recurrence kernel with its panel loads switched off: 4.35e6 requests for the 557 MB it
must stream.  Writes: 5.0e5 requests for the 32 MB of y and r.)  Infinity-Cache hits are counted, not excluded.
The result is merged by hand into profiles/pmc/traffic.json, which bench.py reads for `roofline.traffic`."""
import argparse, collections, csv, glob, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=1_000_000)
ap.add_argument("--dims", type=int, default=50)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "traffic.json"))
args = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="pmc_traffic_", dir="/tmp")
env = dict(os.environ, TMPDIR="/tmp")
cmd = ["rocprofv3", "--kernel-trace", "--pmc", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "--output-format", "csv", "-d", tmp, "-o", "pmc", "--",
       sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--no-host-input", "--no-extra",
       "--cells", str(args.cells), "--dims", str(args.dims)]
subprocess.run(cmd, check=True, env=env, cwd=ROOT, stdout=subprocess.DEVNULL)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(tmp + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for tag, pat in (("knn16_topk", "knn16_topk_kernel<4, 0, 1"), ("knn16_partial_filter", "knn16_partial_filter_kernel"), ("pt_step", "pt_step_kernel<2,"), ("pt_step_p1", "pt_step_kernel<1,"),
                         ("cheby_step", "cheby_step_kernel<2")):
            if pat in k:
                acc[tag][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[(tag, r["Counter_Name"])] += 1
out = {}
for tag, d in acc.items():
    rd = d.get("TCC_EA0_RDREQ_sum", 0.0) / max(cnt[(tag, "TCC_EA0_RDREQ_sum")], 1)
    wr = d.get("TCC_EA0_WRREQ_sum", 0.0) / max(cnt[(tag, "TCC_EA0_WRREQ_sum")], 1)
    out["{}@{}x{}".format(tag, args.cells, args.dims)] = {
        "bytes_per_launch": rd * 128.0 + wr * 64.0,
        "rdreq_per_launch": rd, "wrreq_per_launch": wr, "launches_averaged": cnt[(tag, "TCC_EA0_RDREQ_sum")],
        "note": "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum (own pass, tools/pmc_traffic.py): RDREQ x 128 B + WRREQ x 64 B, "
                "fabric side of L2 (Infinity-Cache hits included)",
    }
os.makedirs(os.path.dirname(args.out), exist_ok=True)
prev = {}
if os.path.exists(args.out):
    prev = json.load(open(args.out))
prev.update(out)
json.dump(prev, open(args.out, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
