set -o pipefail
for i in 1 2; do
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms']['median']); print(' '.join(f\"{r['op'][0]}{r['c_in']}={r['avg_us']}\" for r in d['layers']))"
done
