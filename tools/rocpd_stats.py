#!/usr/bin/env python
"""Per-kernel summary (calls, total / avg / min / max duration) from a rocprofv3 rocpd SQLite file
(`rocprofv3 --kernel-trace --stats` writes <name>_results.db on ROCm 7.2).  Usage:
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    q = """select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc"""
    rows = db.execute(q).fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("%-64s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for name, n, t, a, mn, mx in rows:
        print("%-64s %8d %12.1f %10.2f %10.2f %10.2f %6.1f" % (name[:64], n, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
    span = db.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    print("# kernel time total %.1f us over a %.1f us span" % (tot / 1e3, (span[1] - span[0]) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
