#!/usr/bin/env python3
"""GPU-busy fraction of the timed region from a rocprofv3 --kernel-trace database: union of kernel intervals vs wall span,
per training step (steps are split at the optimizer kernel).  usage: gpu_busy.py <results.db>"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
steps, begin = [], None
for name, s, e in rows:
    if begin is None:
        begin = s
    if "sgd_clip_kernel" in name:
        steps.append((begin, e))
        begin = None
print("steps found:", len(steps))
for i, (b, e) in enumerate(steps[2:], 2):
    ivs = [(s, t) for _, s, t in rows if s >= b and t <= e]
    ivs.sort()
    busy, cur_s, cur_e = 0, None, None
    for s, t in ivs:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, t
        else:
            cur_e = max(cur_e, t)
    busy += cur_e - cur_s
    total = sum(t - s for s, t in ivs)
    gaps = sorted(((ivs[k + 1][0] - max(x[1] for x in ivs[:k + 1])) for k in range(len(ivs) - 1)), reverse=True)
    print("step %d: span %.3f ms  busy(union) %.3f ms  idle %.3f ms  sum of kernel times %.3f ms  kernels %d" % (
        i, (e - b) / 1e6, busy / 1e6, (e - b - busy) / 1e6, total / 1e6, len(ivs)))

# where the idle time sits (last step): gaps by the kernel that follows them
b, e = steps[-1]
ivs = sorted((s, t, name) for name, s, t in rows if s >= b and t <= e)
gaps, reach = [], ivs[0][1]
for k in range(1, len(ivs)):
    s, t, name = ivs[k]
    if s > reach:
        gaps.append((s - reach, ivs[k - 1][2][:48], name[:48], (s - b) / 1e6))
    reach = max(reach, t)
gaps.sort(reverse=True)
print("largest gaps of the last step (us, kernel before -> kernel after, ms into the step):")
for g in gaps[:14]:
    print("  %7.1f  %-48s -> %-48s @%.2f" % (g[0] / 1e3, g[1], g[2], g[3]))
import collections
by = collections.Counter()
for g in gaps:
    by[g[2]] += g[0]
print("idle time by following kernel (us):", [(k, round(v / 1e3, 1)) for k, v in by.most_common(10)])
