#!/usr/bin/env python
"""What the device does between two solves of the bench loop (rocprofv3 kernel trace, rocpd SQLite): every dispatch from the last
deciding k_final of one solve to the first k_chain_init of the next, times relative to that k_final's end.
Usage: python tools/solve_boundary.py <results.db> [which boundary from the end]"""
import sqlite3
import sys


def main(path, back=2):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = db.execute("select s.kernel_name, d.start, d.end, d.%s from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                      "on d.kernel_id = s.id order by d.start" % qcol).fetchall()
    resets = [i for i, r in enumerate(rows) if "k_reset_state" in r[0]]
    i = resets[-back]
    # walk back to the last long k_final (a deciding one) before the reset: the empty look-ahead pass's kernels are short
    a = i
    n_final = 0
    while a > 0 and n_final < 2:
        a -= 1
        if "k_final" in rows[a][0]:
            n_final += 1
    b = i
    while b < len(rows) and "k_chain_init" not in rows[b][0]:
        b += 1
    t0 = rows[a][2]
    print("%-44s %9s %9s %9s  %s" % ("kernel", "start_us", "dur_us", "end_us", "queue"))
    for name, st, en, q in rows[a:b + 2]:
        short = name.split("(")[0].replace("vc::", "")[:44]
        print("%-44s %9.1f %9.1f %9.1f  %s" % (short, (st - t0) / 1e3, (en - st) / 1e3, (en - t0) / 1e3, q))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
