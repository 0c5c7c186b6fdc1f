# round-6 evidence: every line / summary that goes to profiles/r06_* comes from this script (one gpurun call: `bash tools/collect_evidence_r6.sh`)
set -x
export TMPDIR=/tmp
O=gpurun_out/fin6; mkdir -p $O
NB="--no-eager-baseline --no-cpu-baseline"
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_line.err
timeout 300 python bench.py $NB --dtype bf16 > $O/bench_autopet128_bf16.json 2>/dev/null
timeout 300 python bench.py $NB --workload brats128 > $O/bench_brats128_f32.json 2>/dev/null
timeout 300 python bench.py $NB --workload brats128 --dtype bf16 > $O/bench_brats128_bf16.json 2>/dev/null
VELOXSEG_BF16_STORAGE=0 timeout 300 python bench.py $NB --workload brats128 --dtype bf16 > $O/bench_brats128_bf16_operands_only.json 2>/dev/null
timeout 300 python bench.py $NB --workload brats128 --batch 4 > $O/bench_brats128_b4_f32.json 2>/dev/null
timeout 300 python bench.py $NB --workload autopet96 > $O/bench_autopet96.json 2>/dev/null
timeout 300 python bench.py $NB --workload brats96 > $O/bench_brats96.json 2>/dev/null
timeout 300 python bench.py $NB --workload hecktor > $O/bench_hecktor.json 2>/dev/null
# the reference's published GPU protocol (speed_test.py: 10 s + 60 s) and BASELINE configs[4]
timeout 400 python bench.py --mode eval > $O/bench_eval_autopet96_f32.json 2>/dev/null
timeout 300 python bench.py --mode eval --dtype bf16 --t1 20 --no-cpu-baseline > $O/bench_eval_autopet96_bf16.json 2>/dev/null
timeout 300 python bench.py --mode sliding --roi 128 > $O/bench_sliding_roi128.json 2>/dev/null
timeout 300 python bench.py --mode sliding --roi 96 --no-cpu-baseline > $O/bench_sliding_roi96.json 2>/dev/null
# kernel summaries of the benched schedule (the JSON line of the PROFILED process is the sidecar of each CSV)
PR="--no-eager-baseline --no-cpu-baseline --no-kernel-pass --dispersion-steps 0 --steps 100 --warmup 10"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o st --output-format csv -- python3 bench.py $PR > $O/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats96 -o st --output-format csv -- python3 bench.py $PR --workload autopet96 > $O/stats96.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/statsbr -o st --output-format csv -- python3 bench.py $PR --workload brats128 --dtype bf16 > $O/statsbr.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/statshk -o st --output-format csv -- python3 bench.py $PR --workload hecktor > $O/statshk.log 2>&1
# HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes; flag kernels off: under --pmc kernels run one at a time)
PM="--steps 3 --warmup 2 --no-eager-baseline --no-cpu-baseline --dispersion-steps 0 --no-kernel-pass"
for cfg in "autopet128 f32 4" "autopet128 bf16 4" "brats128 f32 2" "brats128 bf16 2"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    VELOXSEG_TAPE_FLAGS=0 timeout 600 rocprofv3 --pmc $c -d $O/pmc_$1_$2_$c -o p --output-format csv -- python3 bench.py $PM --workload $1 --dtype $2 > $O/pmc_$1_$2_$c.log 2>&1
  done
  python tools/pmc_traffic.py $(find $O/pmc_$1_$2_FETCH_SIZE -name '*counter_collection.csv') $(find $O/pmc_$1_$2_WRITE_SIZE -name '*counter_collection.csv') $O/pmc_traffic_$1_$2.json --workload $1 --batch $3 --dtype $2
done
find $O -name '*counter_collection.csv' -delete
find $O -name '*kernel_trace.csv' -delete
timeout 500 python tools/comm_world1_nccl.py 2> $O/comm.err | grep "^RESULT" | sed "s/^RESULT //" > $O/comm_world1_nccl.json
for f in $O/bench_*.json; do echo $f $(tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d.get('ms_per_step'))"); done
for f in stats stats96 statsbr statshk; do echo $f $(tail -1 $O/$f.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config'].get('lanes_on_distinct_hw_queues'), d['config'].get('lane_calibration_spin_us'))"); done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/fin6/pmc_traffic_*.json')):
    d=json.load(open(f)); print(f, d['passes_in_trace'], d['counter_bytes_per_pass'])
PY
