export TMPDIR=/tmp
for q in 4 8; do for x in 2 3; do
echo queues=$q inflight=$x sliding128 $(GPU_MAX_HW_QUEUES=$q VELOXSEG_SW_INFLIGHT=$x python bench.py --mode sliding --roi 128 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done; done
NB="--no-eager-baseline --no-cpu-baseline --no-kernel-pass --dispersion-steps 0"
for q in 4 8; do
echo queues=$q train $(GPU_MAX_HW_QUEUES=$q python bench.py $NB 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('lanes_on_distinct_hw_queues'))")
done
