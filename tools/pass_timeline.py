#!/usr/bin/env python
"""Timeline of one steady-state LM pass from a rocprofv3 kernel trace (rocpd SQLite): every dispatch between two consecutive
k_final launches, with its start relative to the end of the previous k_final, duration and queue.  Shows where the critical path of
the two-stream pass goes (gaps, cross-stream waits).  Usage:  python tools/pass_timeline.py <results.db> [pass index from the end] [marker kernel: k_final | k_trial]"""
import sqlite3
import sys


def main(path, back=5, marker="k_final"):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = db.execute("select s.kernel_name, d.start, d.end, d.%s from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                      "on d.kernel_id = s.id order by d.start" % qcol).fetchall()
    finals = [i for i, r in enumerate(rows) if marker in r[0] and (marker != "k_final" or "merged" not in r[0])]
    a, b = finals[-back - 1], finals[-back]
    t0 = rows[a][2]
    print("%-44s %9s %9s %9s  %s" % ("kernel", "start_us", "dur_us", "end_us", "queue"))
    for name, st, en, q in rows[a + 1:b + 1]:
        short = name.split("(")[0].replace("vc::", "")[:44]
        print("%-44s %9.1f %9.1f %9.1f  %s" % (short, (st - t0) / 1e3, (en - st) / 1e3, (en - t0) / 1e3, q))
    print("# pass length (end of k_final to end of k_final): %.1f us" % ((rows[b][2] - t0) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5, sys.argv[3] if len(sys.argv) > 3 else "k_final")
