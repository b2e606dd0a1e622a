#!/usr/bin/env python3
"""Summarises the kernel dispatches of a rocprofv3 --kernel-trace run whose output is a rocpd SQLite database
(rocprofv3 of ROCm 7.2 writes <name>_results.db unless --output-format csv is given).
usage: python tools/rocpd_kernels.py results.db [--csv out.csv] [--sequence-from k_pip_points --group 6 --skip 1]
Default: one row per kernel name -- calls, total / mean / min / max duration in ms -- the `--stats` table.
--sequence-from K: additionally splits the trace into call sequences that start with kernel K and prints the mean
duration of every kernel per sequence group (tools/msm_bench.py runs 1 + reps calls per size)."""
import argparse
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("bpp::", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--csv")
    ap.add_argument("--sequence-from")
    ap.add_argument("--group", type=int, default=6)
    ap.add_argument("--skip", type=int, default=1)
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    rows = list(cur.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, lds_size from kernels order by start"))
    agg = {}
    for name, st, en, gx, wx, vg, lds in rows:
        d = agg.setdefault(short(name), [])
        d.append((en - st) / 1e6)
    table = sorted(((k, len(v), sum(v), sum(v) / len(v), min(v), max(v)) for k, v in agg.items()), key=lambda r: -r[2])
    tot = sum(r[2] for r in table)
    out = [("kernel", "calls", "total_ms", "mean_ms", "min_ms", "max_ms", "percent")]
    for k, n, t, m, lo, hi in table:
        out.append((k, n, "%.4f" % t, "%.4f" % m, "%.4f" % lo, "%.4f" % hi, "%.2f" % (100 * t / tot)))
    w = csv.writer(open(a.csv, "w", newline="") if a.csv else sys.stdout)
    w.writerows(out)
    if a.sequence_from:
        seqs, cs = [], None
        for name, st, en, gx, wx, vg, lds in rows:
            s = short(name)
            if re.sub(r"<.*", "", s) == a.sequence_from:
                cs = []
                seqs.append(cs)
            if cs is not None:
                cs.append((re.sub(r"<.*", "", s), (en - st) / 1e6, st, en, gx))
        for gi in range(0, len(seqs), a.group):
            grp = seqs[gi:gi + a.group][a.skip:]
            if not grp:
                continue
            per = {}
            for sq in grp:
                for k, d, st, en, gx in sq:
                    per.setdefault(k, []).append(d)
            wall = sum((sq[-1][3] - sq[0][2]) / 1e6 for sq in grp) / len(grp)
            print("# sequence group %d: %d calls, wall %.3f ms (first dispatch start .. last dispatch end)" % (gi // a.group, len(grp), wall))
            for k, v in per.items():
                print("#   %-22s %9.4f ms  x%d" % (k, sum(v) / len(grp), len(v) // len(grp)))


if __name__ == "__main__":
    main()
