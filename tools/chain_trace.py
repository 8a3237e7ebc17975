#!/usr/bin/env python3
"""One training step of a rocprofv3 --kernel-trace database, launch by launch: for the step before the last optimizer kernel, every
dispatch in start order with its stream (M = the caller's, S = the weight-gradient side stream), start offset, duration, the gap to the
previous launch of the same stream and its grid -- the picture of the critical path (the data-gradient chain) that per-kernel totals hide.
Followed by per-level totals of the caller's stream (level = the launch's pixel grid where the kernel's grid shows it).
usage: chain_trace.py <results.db> [step-from-the-end, default 1]"""
import collections
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
qcol = next((c for c in ("stream_id", "queue_id", "queue") if c in cols), None)
rows = cur.execute("select name, start, end, %s, grid_x, grid_y, grid_z, workgroup_x from kernels order by start" % (qcol or "0")).fetchall()
steps, acc = [], []
for r in rows:
    acc.append(r)
    if "sgd_clip_kernel" in r[0]:
        steps.append(acc)
        acc = []
st = steps[-back]
main = st[0][3]
t0 = st[0][1]


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("endo::", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:58]


last_end = {}
print("# step with %d dispatches, span %.3f ms" % (len(st), (st[-1][2] - t0) / 1e6))
print("%-4s %-2s %9s %8s %7s  %-58s %s" % ("#", "st", "start_us", "dur_us", "gap_us", "kernel", "grid"))
gaps = collections.defaultdict(float)
busy = collections.defaultdict(float)
for i, r in enumerate(st):
    s = "M" if r[3] == main else "S"
    gap = (r[1] - last_end[s]) / 1e3 if s in last_end else 0.0
    last_end[s] = max(last_end.get(s, 0), r[2])
    if gap > 0:
        gaps[s] += gap
    busy[s] += (r[2] - r[1]) / 1e3
    print("%-4d %-2s %9.1f %8.1f %7.1f  %-58s %d,%d,%d/%d" % (i, s, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, short(r[0]), r[4], r[5], r[6], r[7]))
for s in sorted(busy):
    print("# stream %s: busy %.3f ms, gaps between its launches %.3f ms" % (s, busy[s] / 1e3, gaps[s] / 1e3))
