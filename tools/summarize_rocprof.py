#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) into the per-kernel text
summary committed under profiles/.   usage: summarize_rocprof.py <results.db> <out.txt> [note] [steps in the profiled run]
With a step count it also writes <out>.json: dispatches per step (all kernels / the library's own), tied to the kernel sources by
their hash (tools/source_id.py) -- bench.py's `dispatches_per_step` reads the newest such file and refuses a stale one."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_id import source_id          # noqa: E402


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
    if len(sys.argv) > 4:
        steps = int(sys.argv[4])
        # per training step: the dispatches between two optimizer kernels (sgd_clip_kernel closes a step), median over the steps
        # after the first two -- bench.py's micro-benchmarks after the timed region do not count
        seq = cur.execute("select name, start, end from kernels order by start").fetchall()
        per_step, n_all, n_own, t_sum = [], 0, 0, 0
        for name, s0, e0 in seq:
            n_all += 1
            n_own += "endo::" in name
            t_sum += e0 - s0
            if "sgd_clip_kernel" in name:
                per_step.append((n_all, n_own, t_sum))
                n_all, n_own, t_sum = 0, 0, 0
        per_step = sorted(per_step[2:]) or [(0, 0, 0)]
        mid = per_step[len(per_step) // 2]
        doc = {"command": note, "steps_in_the_profiled_run": steps, "steps_found": len(per_step) + 2, "source": source_id(),
               "dispatches_per_step": mid[0], "library_kernel_dispatches_per_step": mid[1],
               "kernel_time_ms_per_step": mid[2] / 1e6,
               "note": "dispatches between two optimizer kernels, median over the steps after the first two (memsets / copies "
                       "included in dispatches_per_step; library = kernels of namespace endo)"}
        with open(os.path.splitext(out_path)[0] + ".json", "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)


def by_grid():
    """usage: summarize_rocprof.py --by-grid <results.db> <out.txt>: one row per (kernel, grid) so that each
    resolution level of the network shows up separately."""
    db_path, out_path = sys.argv[2], sys.argv[3]
    cur = sqlite3.connect(db_path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    gx = [c for c in ("grid_x", "grid_size_x") if c in cols]
    gy = [c for c in ("grid_y", "grid_size_y") if c in cols]
    gz = [c for c in ("grid_z", "grid_size_z") if c in cols]
    wx = [c for c in ("workgroup_x", "workgroup_size_x") if c in cols]
    if not gx:
        raise SystemExit("no grid columns in %s" % cols)
    key = "%s, %s, %s, %s" % (gx[0], gy[0], gz[0], wx[0])
    rows = cur.execute("select name, %s, count(*), sum(end-start), avg(end-start) from kernels group by name, %s order by 7 desc" % (key, key)).fetchall()
    total = sum(r[6] for r in rows)
    with open(out_path, "w") as f:
        f.write("# per (kernel, grid) summary; grid in work-items; total %.3f ms\n" % (total / 1e6))
        f.write("%-90s %22s %6s %10s %6s %10s\n" % ("kernel", "grid(x,y,z)/wg", "calls", "total_ms", "pct", "avg_us"))
        for r in rows:
            f.write("%-90s %22s %6d %10.3f %6.2f %10.1f\n" % (r[0][:90], "%d,%d,%d/%d" % (r[1], r[2], r[3], r[4]), r[5], r[6] / 1e6, 100.0 * r[6] / total, r[7] / 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--by-grid":
        by_grid()
    else:
        main()
