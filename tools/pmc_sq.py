#!/usr/bin/env python3
"""Per-kernel wave-time breakdown from one rocprofv3 SQ counter pass (csv): where do the waves of each kernel spend their cycles?
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \\
            --output-format csv -d out -- python3 bench.py --steps 3 --warmup 2 --no-eager-baseline --no-cpu-baseline
  python tools/pmc_sq.py out/*/*counter_collection.csv [name-filter ...]"""
import csv, re, sys, collections
path, filters = sys.argv[1], sys.argv[2:] or ["vx_"]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for row in csv.DictReader(open(path, newline="")):
    name = re.sub(r"\(.*", "", re.sub(r"^void ", "", row["Kernel_Name"]))[:48] + " g" + row["Grid_Size"]
    if not any(f in name for f in filters):
        continue
    acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_WAVE_CYCLES":
        cnt[name] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:40]
print(f"{'kernel':62s} {'n':>4s} {'waveMcyc':>9s} {'wait%':>6s} {'stall%':>7s} {'act%':>6s} {'valu%':>6s} {'lds%':>6s} {'ldsstall%':>9s} {'bankconf':>9s}")
for name, c in rows:
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    pct = lambda k: 100.0 * c.get(k, 0.0) / wc
    print(f"{name:62s} {cnt[name]:4d} {wc / max(cnt[name], 1) / 1e6:9.2f} {pct('SQ_WAIT_ANY'):6.1f} {pct('SQ_WAIT_INST_ANY'):7.1f} {pct('SQ_ACTIVE_INST_ANY'):6.1f} "
          f"{pct('SQ_ACTIVE_INST_VALU'):6.1f} {pct('SQ_ACTIVE_INST_LDS'):6.1f} {pct('SQ_WAIT_INST_LDS'):9.1f} {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(cnt[name], 1) / 1e6:9.2f}")
