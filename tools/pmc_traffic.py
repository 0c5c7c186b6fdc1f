"""Summarise two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE, collected separately as MI355X_MICROARCH.md prescribes) into
profiles/<tag>_pmc_traffic.json: per kernel (template arguments kept, parameter list dropped) the average KB per launch and the corrected
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-byte read requests at 64 bytes).

  python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [name-filter ...]
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    depth, out = 0, []
    for ch in name:                      # cut at the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def collect(path, counter, by_grid=False):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            key = short(row["Kernel_Name"])
            a = acc[(key, int(row["Grid_Size"])) if by_grid else key]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    filters = sys.argv[4:] or ["vx_"]
    f, w = collect(fetch, "FETCH_SIZE"), collect(write, "WRITE_SIZE")
    kernels = {}
    for k in sorted(f, key=lambda k: -f[k][1]):
        if not any(s in k for s in filters) or k not in w:
            continue
        fk, wk = f[k][1] / f[k][0], w[k][1] / w[k][0]
        kernels[k] = {"launches_in_trace": f[k][0], "FETCH_SIZE_KB_avg": round(fk, 1), "WRITE_SIZE_KB_avg": round(wk, 1),
                      "hbm_bytes_per_launch_corrected": int((2 * fk + wk) * 1024)}
    # the same kernel launched with several grid sizes (different shapes): one entry per grid
    fg, wg = collect(fetch, "FETCH_SIZE", True), collect(write, "WRITE_SIZE", True)
    for k in kernels:
        grids = sorted(g for (n, g) in fg if n == k and (n, g) in wg)
        if len(grids) > 1:
            kernels[k]["by_grid"] = {str(g): {"launches_in_trace": fg[(k, g)][0],
                                              "hbm_bytes_per_launch_corrected": int((2 * fg[(k, g)][1] / fg[(k, g)][0] + wg[(k, g)][1] / wg[(k, g)][0]) * 1024)} for g in grids}
    # steps covered by the trace = launches of the once-per-step optimiser kernel
    steps = next((v["launches_in_trace"] for k, v in kernels.items() if k.startswith("vx_adamw_k")), None)
    doc = {"steps_in_trace": steps, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 3 --warmup 2`, B=4 autopet128; traffic = "
                     "2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section), KB -> bytes",
           "kernels": kernels}
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(f"{len(kernels)} kernels -> {out}")


if __name__ == "__main__":
    main()
