#!/usr/bin/env python3
"""Per-kernel statistics per training step from a rocprofv3 results .db: argv: db [top_n]"""
import sqlite3, sys, collections, re
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, end-start from kernels").fetchall()
steps = sum(1 for n, _ in rows if 'adamw' in n)
agg = collections.defaultdict(lambda: [0, 0])
for n, d in rows:
    n = re.sub(r'\(.*', '', n).replace('void ', '')
    agg[n][0] += 1; agg[n][1] += d
tot = sum(v[1] for v in agg.values())
print(f"steps {steps}: kernel time {tot/steps/1e6:.3f} ms/step, launches/step {sum(v[0] for v in agg.values())/steps:.1f}")
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{n[:72]:72s} {v[0]/steps:6.1f} {v[1]/steps/1e3:8.1f} {v[1]/v[0]/1e3:8.1f}")
