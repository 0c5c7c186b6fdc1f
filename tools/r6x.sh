export TMPDIR=/tmp
python -m pytest tests/test_tape_gpu.py tests/test_hip_model_gpu.py tests/test_fused_blocks_gpu.py -x -q 2>&1 | tail -3
NB="--no-eager-baseline --no-cpu-baseline --no-kernel-pass"
for r in 1 2 3; do for x in 0 1; do
echo prefetch=$x autopet128 $(VELOXSEG_WIMG_PREFETCH=$x python bench.py $NB 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['dispersion']['step_ms_p50'])")
done; done
for w in "autopet96 f32" "brats128 f32" "hecktor f32" "brats128 bf16"; do set -- $w; for r in 1 2; do for x in 0 1; do
echo prefetch=$x $1 $2 $(VELOXSEG_WIMG_PREFETCH=$x python bench.py $NB --dispersion-steps 0 --workload $1 --dtype $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done; done; done
