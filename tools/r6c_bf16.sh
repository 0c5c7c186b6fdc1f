set -x
export TMPDIR=/tmp
O=gpurun_out/r6c; mkdir -p $O
NB="--no-eager-baseline --no-cpu-baseline"
timeout 300 python bench.py $NB --workload brats128 > $O/brats128_f32.json 2> $O/err.log
timeout 300 python bench.py $NB --workload brats128 --dtype bf16 > $O/brats128_bf16.json 2>> $O/err.log
VELOXSEG_BF16_STORAGE=0 timeout 300 python bench.py $NB --workload brats128 --dtype bf16 > $O/brats128_bf16_operands_only.json 2>> $O/err.log
timeout 300 python bench.py $NB --dtype bf16 > $O/autopet128_bf16.json 2>> $O/err.log
VELOXSEG_BF16_STORAGE=0 timeout 300 python bench.py $NB --dtype bf16 > $O/autopet128_bf16_operands_only.json 2>> $O/err.log
PM="--steps 3 --warmup 2 --no-eager-baseline --no-cpu-baseline --dispersion-steps 0 --no-kernel-pass"
for cfg in "brats128 f32 2" "brats128 bf16 2" "autopet128 bf16 4" "autopet128 f32 4"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    VELOXSEG_TAPE_FLAGS=0 timeout 600 rocprofv3 --pmc $c -d $O/pmc_$1_$2_$c -o p --output-format csv -- python3 bench.py $PM --workload $1 --dtype $2 > $O/pmc_$1_$2_$c.log 2>&1
  done
  python tools/pmc_traffic.py $(find $O/pmc_$1_$2_FETCH_SIZE -name '*counter_collection.csv') $(find $O/pmc_$1_$2_WRITE_SIZE -name '*counter_collection.csv') $O/pmc_traffic_$1_$2.json --workload $1 --batch $3 --dtype $2
done
find $O -name '*counter_collection.csv' -delete
for f in brats128_f32 brats128_bf16 brats128_bf16_operands_only autopet128_bf16 autopet128_bf16_operands_only; do echo $f; tail -1 $O/$f.json | cut -c1-200; done
tail -5 $O/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6c/pmc_traffic_*.json')):
    d=json.load(open(f)); print(f, d['passes_in_trace'], d['counter_bytes_per_pass'])
PY
