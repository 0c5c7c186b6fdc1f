export TMPDIR=/tmp
python -m pytest tests/test_infer_gpu.py tests/test_tape_gpu.py -x -q 2>&1 | tail -4
for x in 0 1; do
echo keep=$x eval $(VELOXSEG_PREDICTOR_KEEP_IMAGES=$x python bench.py --mode eval --t0 3 --t1 12 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
echo keep=$x eval_bf16 $(VELOXSEG_PREDICTOR_KEEP_IMAGES=$x python bench.py --mode eval --dtype bf16 --t0 3 --t1 12 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done
