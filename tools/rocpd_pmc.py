#!/usr/bin/env python
"""Per-kernel average of each PMC counter from a rocprofv3 rocpd SQLite file (rocprofv3 --pmc X -- cmd).
Usage: python tools/rocpd_pmc.py file_results.db"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(rocpd_pmc_event)")]
    print("# pmc_event columns:", cols)
    pcols = [r[1] for r in db.execute("pragma table_info(rocpd_info_pmc)")]
    print("# info_pmc columns:", pcols)
    q = """select s.kernel_name, p.name, count(*), avg(e.value), sum(e.value)
           from rocpd_pmc_event e
           join rocpd_info_pmc p on e.pmc_id = p.id
           join rocpd_kernel_dispatch d on e.event_id = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name, p.name order by 5 desc"""
    print("%-56s %-14s %8s %16s" % ("kernel", "counter", "calls", "avg_value"))
    for name, cname, n, avg, tot in db.execute(q):
        print("%-56s %-14s %8d %16.1f" % (name[:56], cname, n, avg))


if __name__ == "__main__":
    main(sys.argv[1])
