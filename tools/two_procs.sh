#!/bin/bash
# Two independent training processes sharing ONE GPU: the capture self-check of each (replay == eager pass) must hold under contention.
# This is how the unsynchronised hipMemset of the hop flags was found (first replay wrong in ~1 of 4 runs).
n=${1:-6}
for i in $(seq 1 $n); do
  (timeout 600 python bench.py --steps 6 --warmup 2 --no-eager-baseline --no-cpu-baseline --no-kernel-pass > gpurun_out/tpA$i.txt 2>&1) &
  (timeout 600 python bench.py --steps 6 --warmup 2 --no-eager-baseline --no-cpu-baseline --no-kernel-pass > gpurun_out/tpB$i.txt 2>&1); wait
done
echo "self-check failures: $(cat gpurun_out/tp[AB]*.txt | grep -c 'does not reproduce') of $((2 * n)) processes; finished: $(cat gpurun_out/tp[AB]*.txt | grep -c '^{')"
