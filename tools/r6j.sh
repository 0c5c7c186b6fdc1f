export TMPDIR=/tmp
NB="--no-eager-baseline --no-cpu-baseline --no-kernel-pass"
for r in 1 2; do for x in 0 1 2 3; do
echo xcd=$x autopet128 $(VELOXSEG_ATTN_XCD=$x python bench.py $NB 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['dispersion']['step_ms_p50'])")
done; done
for x in 0 3; do
echo xcd=$x autopet96 $(VELOXSEG_ATTN_XCD=$x python bench.py $NB --dispersion-steps 0 --workload autopet96 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
echo xcd=$x brats128 $(VELOXSEG_ATTN_XCD=$x python bench.py $NB --dispersion-steps 0 --workload brats128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done
