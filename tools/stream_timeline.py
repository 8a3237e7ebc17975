#!/usr/bin/env python3
"""Where a training step's wall time goes, from a rocprofv3 --kernel-trace database: per step (split at the optimizer kernel) the
forward / loss-head / backward / optimizer spans, and inside the backward span the busy and idle time of the caller's stream (the
data-gradient chain) and of the weight-gradient side stream, and how long the join at the end waits for the side stream.
usage: stream_timeline.py <results.db>"""
import collections
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
qcol = next((c for c in ("stream_id", "queue_id", "queue") if c in cols), None)
print("kernels columns:", cols)
rows = cur.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")).fetchall()
steps, begin, acc = [], None, []
for r in rows:
    acc.append(r)
    if "sgd_clip_kernel" in r[0]:
        steps.append(acc)
        acc = []
print("steps:", len(steps), "stream column:", qcol)


def union(ivs):
    ivs = sorted(ivs)
    busy, cs, ce = 0, None, None
    for s, e in ivs:
        if ce is None or s > ce:
            if ce is not None:
                busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return busy + ((ce - cs) if ce is not None else 0)


for i, st in enumerate(steps):
    if i < 2 or len(st) < 50:
        continue
    t0, t1 = st[0][1], st[-1][2]
    first_bwd = next((k for k, r in enumerate(st) if "final_bwd" in r[0]), None)
    head0 = next((k for k, r in enumerate(st) if "depth_scale_pass1" in r[0]), None)
    if first_bwd is None or head0 is None:
        continue
    fwd_end = st[head0][1]
    bwd0 = st[first_bwd][1]
    bwd = [r for r in st[first_bwd:] if "sgd_clip" not in r[0] and "sq_norm" not in r[0]]
    streams = collections.Counter(r[3] for r in bwd)
    main = streams.most_common()[0][0] if False else st[0][3]          # the forward's stream is the caller's stream
    m = [(r[1], r[2]) for r in bwd if r[3] == main]
    s = [(r[1], r[2]) for r in bwd if r[3] != main]
    bwd_end = max(e for _, e in m + s)
    main_end = max(e for _, e in m)
    side_end = max([e for _, e in s] or [bwd0])
    print("step %d: span %.3f ms | forward %.3f | loss head + guard %.3f | backward %.3f | clip+SGD %.3f" % (
        i, (t1 - t0) / 1e6, (fwd_end - t0) / 1e6, (bwd0 - fwd_end) / 1e6, (bwd_end - bwd0) / 1e6, (t1 - bwd_end) / 1e6))
    print("    backward: caller's stream busy %.3f ms, idle %.3f ms, ends at +%.3f | side stream busy %.3f ms, ends at +%.3f (join wait %.3f) | both busy %.3f" % (
        union(m) / 1e6, (main_end - bwd0 - union(m)) / 1e6, (main_end - bwd0) / 1e6, union(s) / 1e6, (side_end - bwd0) / 1e6,
        max(0, side_end - main_end) / 1e6, (union(m) + union(s) - union(m + s)) / 1e6))
    if i == len(steps) - 1:
        by = collections.defaultdict(float)
        for r in bwd:
            if r[3] == main:
                by[r[0][:70]] += (r[2] - r[1]) / 1e6
        print("    caller's stream, backward, by kernel (ms):")
        for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:14]:
            print("       %7.3f  %s" % (v, k))
        fw = collections.defaultdict(float)
        for r in st[:head0]:
            fw[r[0][:70]] += (r[2] - r[1]) / 1e6
        print("    forward, by kernel (ms):")
        for k, v in sorted(fw.items(), key=lambda kv: -kv[1])[:10]:
            print("       %7.3f  %s" % (v, k))
