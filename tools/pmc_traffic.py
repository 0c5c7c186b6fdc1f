"""Summarise two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE, collected separately as MI355X_MICROARCH.md prescribes) into
profiles/<tag>_pmc_traffic.json: per kernel (template arguments kept, parameter list dropped) the average KB per launch and the corrected
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-byte read requests at 64 bytes).

  python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [--workload W --batch B --dtype f32|bf16] [name-filter ...]
(the workload / batch / dtype of the traced bench.py run are recorded in the JSON: bench.py attaches a file's figures only to a line of the same workload, batch and dtype)
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
    rest = sys.argv[4:]
    meta = {"workload": "autopet128", "batch": 4, "dtype": "f32"}
    while rest and rest[0].startswith("--"):
        meta[rest[0][2:]] = int(rest[1]) if rest[0] == "--batch" else rest[1]
        rest = rest[2:]
    filters = rest or ["vx_"]
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
    # Divisor of the per-step sums = forward/backward PASSES in the trace, not optimiser steps: the trace also holds the warm-up / capture / self-check passes, which
    # run every kernel of the step but no AdamW (round 3 divided by the 37 vx_adamw_k launches of a trace with 44 passes: 9.6 GB "per step" was 8.07 GB).  A pass is
    # counted by the kernels that run exactly once in it (the loss finalize / the stem convolution), and the two must agree.
    once = [v["launches_in_trace"] for k, v in kernels.items() if k.startswith("vx_loss_finalize_k") or k.startswith("vx_conv_mfma_fwd_k<1, false>")]
    passes = once[0] if once and all(o == once[0] for o in once) else None
    adamw = next((v["launches_in_trace"] for k, v in kernels.items() if k.startswith("vx_adamw_k")), None)
    total = sum(float(v["hbm_bytes_per_launch_corrected"]) * v["launches_in_trace"] for v in kernels.values())
    doc = {"workload": meta["workload"], "batch": meta["batch"], "dtype": meta["dtype"], "passes_in_trace": passes, "optimizer_steps_in_trace": adamw,
           "counter_bytes_per_pass": (round(total / passes) if passes else None),
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 3 --warmup 2 --dispersion-steps 0 --no-kernel-pass` (workload / batch / dtype: the fields above); traffic = "
                     "2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section), KB -> bytes; "
                     "per-pass sum = sum over kernels of bytes per launch x launches, divided by the launches of the once-per-pass kernels",
           "kernels": kernels}
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(f"{len(kernels)} kernels -> {out}")


if __name__ == "__main__":
    main()
