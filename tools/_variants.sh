set -o pipefail
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --labels 1 --length 5000 --dtype bf16 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms'])"
done
