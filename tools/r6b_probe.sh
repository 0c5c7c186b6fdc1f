set -x
export TMPDIR=/tmp
O=gpurun_out/r6b; mkdir -p $O
python -m pytest tests/test_tape_gpu.py tests/test_custom_ops.py tests/test_concurrency_gpu.py -x -q 2>&1 | tail -4 > $O/pytest.log
timeout 300 python bench.py --mode eval --t0 3 --t1 10 > $O/eval_f32.json 2> $O/eval_f32.err
timeout 300 python bench.py --mode eval --t0 3 --t1 10 --dtype bf16 --no-cpu-baseline > $O/eval_bf16.json 2> $O/eval_bf16.err
timeout 300 python bench.py --mode sliding --roi 128 --t0 2 --t1 8 > $O/sliding128.json 2> $O/sliding128.err
timeout 300 python bench.py --mode sliding --roi 96 --t0 2 --t1 8 --no-cpu-baseline > $O/sliding96.json 2> $O/sliding96.err
timeout 300 python bench.py --workload hecktor --no-eager-baseline --no-cpu-baseline > $O/hecktor.json 2> $O/hecktor.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-eager-baseline --no-cpu-baseline --no-kernel-pass --dispersion-steps 0 > $O/stats.log 2>&1
find $O -name '*kernel_trace.csv' -delete
tail -3 $O/pytest.log
for f in eval_f32 eval_bf16 sliding128 sliding96 hecktor; do echo $f; tail -1 $O/$f.json | cut -c1-250; tail -2 $O/$f.err; done
tail -1 $O/stats.log | cut -c1-600
