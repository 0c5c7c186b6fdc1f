for a in "autopet96 4" "brats96 2" "autopet128 8" "autopet128 1" "autopet128 2" "brats128 2" "brats128 4" "brats128 1"; do set -- $a; timeout 300 python bench.py --no-eager-baseline --no-cpu-baseline --no-kernel-pass --workload $1 --batch $2 2>&1 | grep -E "^\{|Warn" | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 B$2', d['value'], d['ms_per_step'], d['config']['launch'][:60])
    else: print(l.strip()[:300])
"; done
