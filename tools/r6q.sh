export TMPDIR=/tmp
NB="--no-eager-baseline --no-cpu-baseline --no-kernel-pass"
for w in "brats128 f32" "brats128 bf16"; do set -- $w; for r in 1 2 3; do for x in 0 2; do
echo m1=$x $1 $2 $(VELOXSEG_F16_BWD_M1=$x python bench.py $NB --dispersion-steps 0 --workload $1 --dtype $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done; done; done
