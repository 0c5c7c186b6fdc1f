export TMPDIR=/tmp
NB="--no-eager-baseline --no-cpu-baseline --no-kernel-pass"
for w in "autopet96 f32" "hecktor f32" "autopet128 bf16"; do set -- $w; for r in 1 2 3; do for x in 768 384; do
echo blocks=$x $1 $2 $(VELOXSEG_EXPAND_WG_BLOCKS=$x python bench.py $NB --dispersion-steps 0 --workload $1 --dtype $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done; done; done
