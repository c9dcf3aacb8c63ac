#!/usr/bin/env python
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, as
MI355X_MICROARCH.md prescribes).  Launches that exit early (converged solve, queued passes) are excluded:
"active" = launches whose counter exceeds 5 % of the kernel's maximum.
Usage: python tools/pmc_traffic.py fetch_results.db write_results.db [workload out.json]
With the two extra arguments the per-kernel byte counts are merged into out.json under `workload` (what bench.py reads as
profiles/pmc_traffic_latest.json)."""
import re
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    q = """select s.kernel_name, e.value from rocpd_pmc_event e
           join rocpd_info_pmc p on e.pmc_id = p.id
           join rocpd_kernel_dispatch d on e.event_id = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           where p.name = ?"""
    out = {}
    for name, val in db.execute(q, (counter,)):
        out.setdefault(name, []).append(float(val))
    return out


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)", name)
    return m.group(1) if m else name[:24]


def main(fetch_db, write_db, workload=None, out_json=None):
    F = per_kernel(fetch_db, "FETCH_SIZE"); W = per_kernel(write_db, "WRITE_SIZE")
    print("# FETCH_SIZE / WRITE_SIZE in KB as reported; gfx950 correction: FETCH_SIZE counts 64 B per 128-B request -> x2")
    print("%-20s %9s %9s %14s %14s %16s" % ("kernel", "launches", "active", "FETCH_KB", "WRITE_KB", "bytes(2F+W)"))
    rows = []
    for name in F:
        f = F[name]; w = W.get(name, [0.0])
        fa = [x for x in f if x > 0.05 * max(f)] or [0.0]
        wa = [x for x in w if x > 0.05 * max(w)] or [0.0]
        favg = sum(fa) / len(fa); wavg = sum(wa) / len(wa)
        rows.append((2048.0 * favg + 1024.0 * wavg, short(name), len(f), len(fa), favg, wavg))
    for b, n, nl, na, fa, wa in sorted(rows, reverse=True):
        print("%-20s %9d %9d %14.1f %14.1f %16.0f" % (n, nl, na, fa, wa, b))
    # kernels launched several times per LM pass (the levels of the chain elimination): bytes per PASS = everything the kernel
    # moved, over the number of passes (= active launches of the deciding kernel)
    passes = 0
    for name in F:
        if short(name) in ("k_final", "k_trial"):
            f = F[name]
            passes = max(passes, len([x for x in f if x > 0.05 * max(f)]))
    per_pass = {}
    if passes:
        print("# per LM pass (%d passes): all launches of a kernel summed" % passes)
        for name in F:
            tot = 2048.0 * sum(F[name]) + 1024.0 * sum(W.get(name, [0.0]))
            per_pass[short(name) + "@pass"] = tot / passes
        for n in ("k_chain_fwd", "k_chain_fwd2", "k_chain_back"):
            if n + "@pass" in per_pass:
                print("%-20s %16.0f bytes per pass" % (n, per_pass[n + "@pass"]))
    if workload and out_json:
        import json
        import os
        d = {}
        if os.path.exists(out_json):
            d = json.load(open(out_json))
        d[workload] = {n: b for b, n, nl, na, fa, wa in rows}
        d[workload].update(per_pass)
        json.dump(d, open(out_json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(*sys.argv[1:5])
