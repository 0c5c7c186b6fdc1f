export TMPDIR=/tmp
NB="--no-eager-baseline --no-cpu-baseline --no-kernel-pass"
for w in "brats128 f32" "brats128 bf16" "brats96 f32" "autopet128 f32"; do set -- $w; for r in 1 2; do for x in 0 1; do
echo expandpf=$x $1 $2 $(VELOXSEG_EXPAND_PREFETCH=$x python bench.py $NB --dispersion-steps 0 --workload $1 --dtype $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done; done; done
VELOXSEG_EXPAND_PREFETCH=1 python tools/tape_critical_path.py brats128 2>&1 | head -12 | cut -c1-170
VELOXSEG_EXPAND_PREFETCH=0 python tools/tape_critical_path.py brats128 2>&1 | head -12 | cut -c1-170
