python -m pytest tests -q -k "adam or optim or train_harness or engine" -m gpu 2>&1 | tail -2
for t in 1 2; do python bench.py --no-eager-baseline --no-cpu-baseline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_pass']['total_ms'], d['config'].get('final_loss'))"; done
