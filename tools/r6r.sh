export TMPDIR=/tmp
python -m pytest tests/test_hip_ops_gpu.py -x -q -k "scatter or pwa" 2>&1 | tail -3
python -m pytest tests/test_hip_model_gpu.py tests/test_tape_gpu.py tests/test_pwa_fused_gpu.py -x -q 2>&1 | tail -2
NB="--no-eager-baseline --no-cpu-baseline --no-kernel-pass"
for r in 1 2 3; do for x in 0 1; do
echo w=$x autopet128 $(VELOXSEG_SCATTER_BWD_W=$x python bench.py $NB 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['dispersion']['step_ms_p50'])")
done; done
for w in autopet96 brats128; do for x in 0 1; do
echo w=$x $w $(VELOXSEG_SCATTER_BWD_W=$x python bench.py $NB --dispersion-steps 0 --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done; done
