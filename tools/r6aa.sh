export TMPDIR=/tmp
python -m pytest tests/test_infer_gpu.py -x -q 2>&1 | tail -2
for x in 0 1; do
echo keep=$x sliding128 $(VELOXSEG_PREDICTOR_KEEP_IMAGES=$x python bench.py --mode sliding --roi 128 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
echo keep=$x sliding96 $(VELOXSEG_PREDICTOR_KEEP_IMAGES=$x python bench.py --mode sliding --roi 96 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done
