set -x
O=gpurun_out/r6d; mkdir -p $O
python -m pytest tests/test_hip_ops_gpu.py tests/test_pwa_fused_gpu.py tests/test_hip_model_gpu.py tests/test_bf16_gpu.py -x -q 2>&1 | tail -4 > $O/pytest.log
NB="--no-eager-baseline --no-cpu-baseline --no-kernel-pass --dispersion-steps 0"
for w in brats128 brats96 hecktor autopet96 autopet128; do
  timeout 300 python bench.py $NB --workload $w > $O/${w}_new.json 2>> $O/err.log
  VELOXSEG_B1_SHORT=0 timeout 300 python bench.py $NB --workload $w > $O/${w}_old.json 2>> $O/err.log
done
VELOXSEG_B1_FEW=1 timeout 300 python bench.py $NB > $O/autopet128_few.json 2>> $O/err.log
timeout 300 python bench.py $NB --workload brats128 --dtype bf16 > $O/brats128_bf16_new.json 2>> $O/err.log
tail -3 $O/pytest.log
for f in $O/*.json; do echo $f $(tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); done
