"""Idle gaps of the GPU inside one benchmark step, from a rocprofv3 kernel trace (csv):
   rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python bench.py --cells 1000000 --steps 3 --warmup 1 --cpu-sample 0 --no-host-input
   python tools/step_gaps.py /tmp/tr/*_kernel_trace.csv [min_gap_us]
The last step is cut out at the search kernel's dispatches; busy time = union of the dispatch intervals."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
marks = [i for i, e in enumerate(ev) if "knn16_topk_kernel" in e[2] and ", 1, " in e[2]]  # first-pass search, one per step
if len(marks) < 2:
    sys.exit("fewer than two steps in the trace")
# a step = from the first dispatch after the previous step's last kernel ... approximate by the span between two search kernels
a, b = marks[-2], marks[-1]
seg = ev[a:b]
t0, t1 = seg[0][0], seg[-1][1]
busy, cur_end, gaps = 0, seg[0][0], []
prev = None
for s, e, n in seg:
    if s > cur_end:
        if (s - cur_end) / 1e3 >= min_gap:
            gaps.append(((s - cur_end) / 1e3, prev, n))
        busy += 0
        cur_start = s
    busy += max(0, e - max(s, cur_end))
    if e > cur_end:
        cur_end, prev = e, n
print("span between two search kernels: %.2f ms, GPU busy (union) %.2f ms, idle %.2f ms" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
tot = 0.0
for g, p, n in sorted(gaps, reverse=True)[:25]:
    tot += g
    print("  %7.1f us idle between %-60s and %s" % (g, (p or "")[:60], n[:60]))
print("gaps >= %.0f us: %d, %.2f ms in all" % (min_gap, len(gaps), sum(g for g, _, _ in gaps) / 1e3))
