#!/usr/bin/env python3
"""Timeline statistics from a rocprofv3 kernel_trace.csv of bench.py: per-step span, per-queue busy time, idle gaps, concurrency.
argv: csv [steps_to_analyse_from_the_end]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]) for r in rows))
# steps are delimited by the fused AdamW kernel
ends = [i for i, e in enumerate(ev) if "adamw" in e[3]]
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for a, b in list(zip(ends[:-1], ends[1:]))[-nst:]:
    seg = ev[a + 1:b + 1]
    t0, t1 = seg[0][0], max(e[1] for e in seg)
    span = (t1 - t0) / 1e6
    busy = collections.Counter()
    for s, e, q, n in seg:
        busy[q] += (e - s) / 1e6
    # union of busy intervals (any queue)
    pts = sorted([(s, 1) for s, e, q, n in seg] + [(e, -1) for s, e, q, n in seg])
    act, last, union, conc2 = 0, t0, 0, 0
    for t, d in pts:
        if act > 0:
            union += t - last
        if act > 1:
            conc2 += t - last
        act += d
        last = t
    print(f"step: {len(seg)} kernels, span {span:.2f} ms, kernel-time sum {sum(busy.values()):.2f} ms, GPU non-idle {union/1e6:.2f} ms, >=2 kernels in flight {conc2/1e6:.2f} ms; per queue busy ms: {dict((q, round(v, 2)) for q, v in sorted(busy.items()))}")
# top idle gaps on the main queue of the last step
seg = ev[ends[-2] + 1:ends[-1] + 1]
mainq = collections.Counter(q for _, _, q, _ in seg).most_common(1)[0][0]
m = [e for e in seg if e[2] == mainq]
gaps = sorted(((m[i + 1][0] - m[i][1]) / 1e3, m[i][3][:50], m[i + 1][3][:50]) for i in range(len(m) - 1))
print("largest gaps on the main queue (us, after kernel, before kernel):")
for g in gaps[-12:][::-1]:
    print("  %8.1f  %-50s -> %s" % g)
tot_gap = sum(g[0] for g in gaps if g[0] > 0)
print("sum of positive gaps on the main queue: %.2f ms over %d kernels" % (tot_gap / 1e3, len(m)))
