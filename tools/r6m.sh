export TMPDIR=/tmp
O=gpurun_out/r6m; mkdir -p $O
python -m pytest tests/test_hip_ops_gpu.py -x -q -k "stem or patch or down" 2>&1 | tail -3
python -m pytest tests/test_hip_model_gpu.py tests/test_tape_gpu.py -x -q 2>&1 | tail -2
NB="--no-eager-baseline --no-cpu-baseline --no-kernel-pass"
for r in 1 2 3; do for x in 0 1; do
echo mxfwd=$x autopet128 $(VELOXSEG_STEM_ABSMAX_FWD=$x python bench.py $NB 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['dispersion']['step_ms_p50'])")
done; done
for w in "hecktor f32" "brats128 bf16" "autopet96 f32"; do set -- $w; for x in 0 1; do
echo mxfwd=$x $1 $2 $(VELOXSEG_STEM_ABSMAX_FWD=$x python bench.py $NB --dispersion-steps 0 --workload $1 --dtype $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done; done
PM="--steps 3 --warmup 2 --no-eager-baseline --no-cpu-baseline --dispersion-steps 0 --no-kernel-pass"
for c in FETCH_SIZE WRITE_SIZE; do
  VELOXSEG_TAPE_FLAGS=0 timeout 600 rocprofv3 --pmc $c -d $O/pmc_$c -o p --output-format csv -- python3 bench.py $PM > $O/pmc_$c.log 2>&1
done
python tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv') $(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv') $O/pmc_traffic_autopet128_f32.json --workload autopet128 --batch 4 --dtype f32
find $O -name '*counter_collection.csv' -delete
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6m/pmc_traffic_autopet128_f32.json')); P=d['passes_in_trace']; print(P, d['counter_bytes_per_pass'])
for k,v in d['kernels'].items():
    if 'stem' in k: print(k, v['launches_in_trace']/P, v['hbm_bytes_per_launch_corrected']/1e6)
PY
