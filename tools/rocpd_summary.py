#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` reports: calls, total / average / min / max duration, share.

    python tools/rocpd_summary.py gpurun_out/prof_r1/bench_r1_results.db > profiles/r01_bench_1M_kernel_stats.md
"""
import sqlite3
import sys


def main(path, pmc=False):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc"
    ).fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | lds B | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for name, n, s, a, mn, mx, vg, lds, g, wg in rows:
        short = name if len(name) < 110 else name[:107] + "..."
        print("| `{}` | {} | {:.3f} | {:.1f} | {:.1f} | {:.1f} | {:.2f} | {} | {} | {} | {} |".format(
            short, n, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot, vg, lds, g, wg))
    print()
    print("total kernel time: {:.3f} ms over {} dispatches".format(tot / 1e6, sum(r[1] for r in rows)))


if __name__ == "__main__":
    main(sys.argv[1])
