#!/usr/bin/env python3
"""Print a per-kernel table (calls/step, ms/step, avg us) from a rocprofv3 *_kernel_stats.csv; argv: csv, steps-in-trace"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms/step %.3f" % (tot / n / 1e6))
for r in rows[:top]:
    print("%-86s calls %6.1f  ms %7.3f  avg_us %8.1f" % (r["Name"][:86], int(r["Calls"]) / n, float(r["TotalDurationNs"]) / n / 1e6, float(r["AverageNs"]) / 1e3))
