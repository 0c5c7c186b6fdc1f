export TMPDIR=/tmp
NB="--no-eager-baseline --no-cpu-baseline --no-kernel-pass"
for r in 1 2 3; do for x in 0 1; do
echo fused=$x autopet128 $(VELOXSEG_PATCH_FUSED=$x python bench.py $NB 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['dispersion']['step_ms_p50'])")
done; done
for w in "hecktor f32" "brats128 f32"; do set -- $w; for x in 0 1; do
echo fused=$x $1 $2 $(VELOXSEG_PATCH_FUSED=$x python bench.py $NB --dispersion-steps 0 --workload $1 --dtype $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done; done
