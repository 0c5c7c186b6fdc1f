#!/usr/bin/env python3
"""Copy what `bash tools/collect_evidence_r6.sh` left under gpurun_out/fin6 (scratch) into profiles/r06_* (tracked):
bench lines (the last line of each file must parse as JSON), counter-traffic summaries, the one-rank RCCL line, and every rocprofv3 kernel summary under the name
`r06_kernel_stats_<workload>_profiled_<patches/s of the PROFILED process>_patches_s.csv` with that process's own JSON line beside it (`.json`)."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
O = os.path.join(ROOT, "gpurun_out", "fin6")
P = os.path.join(ROOT, "profiles")


def main() -> int:
    if not os.path.isdir(O):
        print("no gpurun_out/fin6: run tools/collect_evidence_r6.sh on the GPU box first")
        return 1
    for f in glob.glob(os.path.join(P, "r06_*")):
        os.remove(f)
    for f in sorted(glob.glob(os.path.join(O, "bench_*.json"))):
        line = open(f).read().strip().splitlines()[-1]
        d = json.loads(line)
        open(os.path.join(P, "r06_" + os.path.basename(f)), "w").write(line + "\n")
        print(f"{os.path.basename(f):48s} {d['value']:10.3f} {d['unit']}  {d.get('ms_per_step')} ms")
    for f in sorted(glob.glob(os.path.join(O, "pmc_traffic_*.json"))):
        shutil.copy(f, os.path.join(P, "r06_" + os.path.basename(f)))
        d = json.load(open(f))
        print(f"{os.path.basename(f):48s} {d['counter_bytes_per_pass'] / 1e9:.3f} GB per pass ({d['passes_in_trace']} passes in the trace)")
    shutil.copy(os.path.join(O, "comm_world1_nccl.json"), os.path.join(P, "r06_comm_world1_nccl.json"))
    names = {"stats": "autopet128", "stats96": "autopet96", "statsbr": "brats128_bf16", "statshk": "hecktor"}
    for k, w in names.items():
        lines = [l for l in open(os.path.join(O, k + ".log")).read().splitlines() if l.startswith('{"metric"')]
        d = json.loads(lines[-1])
        csvs = glob.glob(os.path.join(O, k, "**", "*kernel_stats.csv"), recursive=True)
        assert len(csvs) == 1, csvs
        base = os.path.join(P, f"r06_kernel_stats_{w}_profiled_{int(round(d['value']))}_patches_s")
        shutil.copy(csvs[0], base + ".csv")
        open(base + ".json", "w").write(lines[-1] + "\n")
        print(f"{os.path.basename(base):64s} queues {d['config'].get('lanes_on_distinct_hw_queues')} spin {d['config'].get('lane_calibration_spin_us')} us")
    return 0


if __name__ == "__main__":
    sys.exit(main())
