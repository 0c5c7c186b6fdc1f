#!/bin/bash
# profiles/r05_tape_soak.txt: 3000 replays (a device synchronisation after every one) and 500 fuzzed replays per workload, default configuration (stem f16 kernel on)
out=${1:-gpurun_out/r05_tape_soak.txt}
mkdir -p $(dirname $out); : > $out
export VX_SOAK_QUIET=1
for wl in autopet128 autopet96 brats128; do
  VX_SYNC_EVERY=1 timeout 900 python tools/tape_soak.py $wl 4 3000 2>&1 | grep -v "^replay\|amdgpu.ids" | tail -1 >> $out
  VX_SOAK_FUZZ=30,0.2 timeout 900 python tools/tape_soak.py $wl 4 500 2>&1 | grep -v "^replay\|amdgpu.ids" | tail -1 >> $out
done
VX_SOAK_SERIAL=24 VX_SOAK_EXHAUSTIVE=1 timeout 900 python tools/tape_soak.py autopet128 4 2 2>&1 | grep "serial audit" >> $out
cat $out
