export TMPDIR=/tmp
python -m pytest tests/test_hip_ops_gpu.py -x -q -k "scatter" 2>&1 | tail -2
NB="--no-eager-baseline --no-cpu-baseline --no-kernel-pass"
for w in "brats128 f32" "brats128 bf16" "brats96 f32" "hecktor f32" "autopet128 bf16"; do set -- $w; for r in 1 2 3; do for x in 0 1; do
echo w=$x $1 $2 $(VELOXSEG_SCATTER_BWD_W=$x python bench.py $NB --dispersion-steps 0 --workload $1 --dtype $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done; done; done
