#!/usr/bin/env python3
"""Summarise a rocprofv3 run (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python3 ...`, which writes a rocpd
SQLite database NAME_results.db): per-kernel totals as CSV (the file committed under profiles/) and, with --timeline N, the
launch sequence of the last N-th of the run (one step when N = number of steps profiled).

    python tools/prof_summary.py gpurun_out/prof/x_results.db --csv profiles/r01_x_kernel_stats.csv [--timeline 6]
"""
import argparse
import csv
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--csv")
    ap.add_argument("--timeline", type=int, default=0)
    ap.add_argument("--top", type=int, default=30)
    a = ap.parse_args()
    c = sqlite3.connect(a.db).cursor()
    rows = list(c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, accum_vgpr_count "
                          "from kernels order by start"))
    agg = {}
    for r in rows:
        e = agg.setdefault(r[0], [0, 0, 1 << 62, 0])
        d = r[2] - r[1]
        e[0] += 1
        e[1] += d
        e[2] = min(e[2], d)
        e[3] = max(e[3], d)
    total = sum(e[1] for e in agg.values()) or 1
    table = sorted(agg.items(), key=lambda kv: -kv[1][1])
    out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")]
    for name, e in table:
        out.append((name, e[0], e[1], round(e[1] / e[0], 1), round(100.0 * e[1] / total, 3), e[2], e[3]))
    if a.csv:
        with open(a.csv, "w", newline="") as f:
            csv.writer(f).writerows(out)
    for r in out[:a.top + 1]:
        print(f"{str(r[0])[:84]:84s} {r[1]:>6} {r[2]:>12} {r[3]:>11} {r[4]:>7}")
    if a.timeline:
        n = len(rows) // a.timeline
        step = rows[-n:]
        t0 = step[0][1]
        print(f"--- last {n} launches ({(step[-1][2] - t0) / 1e3:.1f} us)")
        for r in step:
            print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f} {r[0].split('(')[0][:52]:52s} grid {r[3] // max(r[6], 1)}x{r[4]}x{r[5]}")


if __name__ == "__main__":
    sys.exit(main())
