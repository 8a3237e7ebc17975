#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) into the per-kernel text
summary committed under profiles/.   usage: summarize_rocprof.py <results.db> <out.txt> [note]"""
import sqlite3
import sys


def main():
    db_path, out_path = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "avg(vgpr_count), avg(accum_vgpr_count), avg(lds_size) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(out_path, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (%s)\n" % db_path.split("/")[-1])
        if note:
            f.write("# %s\n" % note)
        f.write("# total kernel time %.3f ms over %d dispatches\n" % (total / 1e6, sum(r[1] for r in rows)))
        f.write("%-100s %7s %11s %6s %10s %10s %10s %5s %5s %7s\n" % ("kernel", "calls", "total_ms", "pct", "avg_us", "min_us", "max_us", "vgpr", "agpr", "lds"))
        for r in rows:
            f.write("%-100s %7d %11.3f %6.2f %10.1f %10.1f %10.1f %5d %5d %7d\n" % (
                r[0][:100], r[1], r[2] / 1e6, 100.0 * r[2] / total, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, r[6] or 0, r[7] or 0, r[8] or 0))


if __name__ == "__main__":
    main()
