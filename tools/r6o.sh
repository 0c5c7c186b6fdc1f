export TMPDIR=/tmp
NB="--no-eager-baseline --no-cpu-baseline --no-kernel-pass"
for r in 1 2 3; do for x in 64 96 128; do
echo blocks=$x autopet128 $(VELOXSEG_WG_TZ_BLOCKS=$x python bench.py $NB 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['dispersion']['step_ms_p50'])")
done; done
