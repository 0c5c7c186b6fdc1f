set -x
mkdir -p gpurun_out/fin
export TMPDIR=/tmp
if [ -z "$VX_EVIDENCE_PROFILES_ONLY" ]; then
timeout 600 python bench.py > gpurun_out/fin/bench_line.json 2> gpurun_out/fin/bench_line.err
timeout 300 python bench.py --no-eager-baseline --no-cpu-baseline --dtype bf16 > gpurun_out/fin/bench_autopet128_bf16.json 2>/dev/null
timeout 300 python bench.py --no-eager-baseline --no-cpu-baseline --workload brats128 > gpurun_out/fin/bench_brats128_f32.json 2>/dev/null
timeout 300 python bench.py --no-eager-baseline --no-cpu-baseline --workload brats128 --dtype bf16 > gpurun_out/fin/bench_brats128_bf16.json 2>/dev/null
timeout 300 python bench.py --no-eager-baseline --no-cpu-baseline --workload brats128 --batch 4 > gpurun_out/fin/bench_brats128_b4_f32.json 2>/dev/null
timeout 300 python bench.py --no-eager-baseline --no-cpu-baseline --workload autopet96 > gpurun_out/fin/bench_autopet96.json 2>/dev/null
timeout 300 python bench.py --no-eager-baseline --no-cpu-baseline --workload brats96 > gpurun_out/fin/bench_brats96.json 2>/dev/null
fi
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/fin/stats96 -o st --output-format csv -- python3 bench.py --workload autopet96 --steps 100 --warmup 10 --no-eager-baseline --no-cpu-baseline --no-kernel-pass --dispersion-steps 0 > gpurun_out/fin/stats96.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/fin/stats -o st --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-eager-baseline --no-cpu-baseline --no-kernel-pass --dispersion-steps 0 > gpurun_out/fin/stats.log 2>&1
# (flag kernels off: under --pmc kernels run one at a time, a polling kernel would never see its flag set)
VELOXSEG_TAPE_FLAGS=0 timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/fin/fetch -o f --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-eager-baseline --no-cpu-baseline --dispersion-steps 0 > gpurun_out/fin/fetch.log 2>&1
VELOXSEG_TAPE_FLAGS=0 timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/fin/write -o w --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-eager-baseline --no-cpu-baseline --dispersion-steps 0 > gpurun_out/fin/write.log 2>&1
python tools/pmc_traffic.py $(find gpurun_out/fin/fetch -name '*counter_collection.csv') $(find gpurun_out/fin/write -name '*counter_collection.csv') gpurun_out/fin/pmc_traffic.json
find gpurun_out/fin -name '*counter_collection.csv' -delete
find gpurun_out/fin -name '*kernel_trace.csv' -delete
ls -la gpurun_out/fin gpurun_out/fin/stats
tail -1 gpurun_out/fin/bench_line.json | cut -c1-300
