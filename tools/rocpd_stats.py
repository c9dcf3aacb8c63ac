#!/usr/bin/env python
"""Per-kernel summary (calls, total / avg / min / max duration) from a rocprofv3 rocpd SQLite file
(`rocprofv3 --kernel-trace --stats` writes <name>_results.db on ROCm 7.2).  Usage:
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    q = """select s.kernel_name, d.end - d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id"""
    per = {}
    for name, dur in db.execute(q):
        per.setdefault(name, []).append(dur)
    rows = sorted(((n, len(v), sum(v), sum(v) / len(v), min(v), max(v), v) for n, v in per.items()), key=lambda r: -r[2])
    tot = sum(r[2] for r in rows) or 1
    # active launches: passes queued past the end of a solve return at their first instruction (a few us) and pull the plain average
    # down -- launches shorter than half the kernel's median are left out of the `active` columns
    print("%-64s %8s %12s %10s %10s %10s %6s %8s %11s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "active", "active_avg"))
    for name, n, t, a, mn, mx, v in rows:
        med = sorted(v)[len(v) // 2]
        act = [x for x in v if x >= 0.5 * med]
        print("%-64s %8d %12.1f %10.2f %10.2f %10.2f %6.1f %8d %11.2f" % (name[:64], n, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot, len(act), sum(act) / len(act) / 1e3))
    span = db.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    print("# kernel time total %.1f us over a %.1f us span; active = launches of at least half the kernel's median duration (early-exit launches excluded)" % (tot / 1e3, (span[1] - span[0]) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
