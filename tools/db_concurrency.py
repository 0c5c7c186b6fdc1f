#!/usr/bin/env python3
"""GPU concurrency of the last training step from a rocprofv3 results .db: how long 0, 1, 2, 3+ kernels were in flight, the longest
idle gaps (with the kernels either side) and the time spent alone by kernel name.  argv: db"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if 'adamw' in r[0]]
seg = rows[ends[-2] + 1:ends[-1] + 1]
ev = []
for i, r in enumerate(seg):
    ev.append((r[1], 1, i)); ev.append((r[2], -1, i))
ev.sort()
depth = 0; last = ev[0][0]; hist = collections.Counter(); alone = collections.Counter(); live = set(); gaps = []
prev_end_kernel = None
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        hist[min(depth, 4)] += dt
        if depth == 1:
            alone[seg[next(iter(live))][0][:70]] += dt
        if depth == 0:
            gaps.append((dt, last, prev_end_kernel, seg[i][0][:60]))
    if d == 1: live.add(i)
    else: live.discard(i); prev_end_kernel = seg[i][0][:60]
    depth += d; last = t
tot = sum(hist.values())
print(f"{len(seg)} kernels, span {tot/1e6:.2f} ms")
for k in sorted(hist): print(f"  {k}{'+' if k == 4 else ' '} kernels in flight: {hist[k]/1e6:6.2f} ms ({100*hist[k]/tot:4.1f} %)")
print(f"idle gaps: {len(gaps)}, mean {sum(g[0] for g in gaps)/max(len(gaps),1)/1e3:.1f} us")
t0 = seg[0][1]
for g in sorted(gaps, reverse=True)[:15]: print(f"  {g[0]/1e3:7.1f} us at {(g[1]-t0)/1e6:6.2f} ms  after {g[2]}  before {g[3]}")
print("time running alone, by kernel:")
for n, v in alone.most_common(25): print(f"  {v/1e3:8.1f} us  {n}")
