set -x
export TMPDIR=/tmp
O=gpurun_out/r6g; mkdir -p $O
python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -2
NB="--no-eager-baseline --no-cpu-baseline --no-kernel-pass --dispersion-steps 0"
for w in brats128 autopet128; do for i in 1 2; do echo $w $(python bench.py $NB --workload $w --dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); done; done
PM="--steps 3 --warmup 2 --no-eager-baseline --no-cpu-baseline --dispersion-steps 0 --no-kernel-pass"
for c in FETCH_SIZE WRITE_SIZE; do
  VELOXSEG_TAPE_FLAGS=0 timeout 600 rocprofv3 --pmc $c -d $O/pmc_$c -o p --output-format csv -- python3 bench.py $PM --workload brats128 --dtype bf16 > $O/pmc_$c.log 2>&1
done
python tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv') $(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv') $O/pmc_traffic_brats128_bf16.json --workload brats128 --batch 2 --dtype bf16
find $O -name '*counter_collection.csv' -delete
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6g/pmc_traffic_brats128_bf16.json')); P=d['passes_in_trace']; print(P, d['counter_bytes_per_pass'])
for k,v in d['kernels'].items():
    if 'expand' in k: print(k, v['launches_in_trace']/P, v['hbm_bytes_per_launch_corrected']/1e6)
PY
