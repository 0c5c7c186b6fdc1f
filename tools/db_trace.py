#!/usr/bin/env python3
"""Ordered kernel trace of the last training step from a rocprofv3 results .db (sqlite): argv: db [out.txt].
Steps are delimited by the fused AdamW kernel."""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, queue_id, stream_id, grid_x, workgroup_x, vgpr_count, lds_size from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if 'adamw' in r[0]]
a, b = ends[-2], ends[-1]
seg = rows[a + 1:b + 1]
t0 = seg[0][1]
span = (max(r[2] for r in seg) - t0) / 1e6
busy = collections.Counter()
for r in seg:
    busy[r[3]] += (r[2] - r[1]) / 1e6
print(f"{len(seg)} kernels, span {span:.2f} ms, kernel-time sum {sum(busy.values()):.2f} ms, per queue {dict((q, round(v, 2)) for q, v in busy.items())}")
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
for r in seg:
    out.write(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-r[1])/1e3:7.1f} q{r[3]} g{r[5]//max(r[6],1)}x{r[6]} v{r[7]} l{r[8]} {r[0][:90]}\n")
