export TMPDIR=/tmp
for x in 2 3 4; do
echo inflight=$x sliding128 $(VELOXSEG_SW_INFLIGHT=$x python bench.py --mode sliding --roi 128 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
echo inflight=$x sliding96 $(VELOXSEG_SW_INFLIGHT=$x python bench.py --mode sliding --roi 96 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done
