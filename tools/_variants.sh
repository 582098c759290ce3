set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -q -x > $O/r02l_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r02l_tests.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/r02l_bench.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02l_bench.json').read().strip().splitlines()[-1])
def show(x, name): print(name[:70], x['value'], x['ms_per_step'], x['step_ms']['median'], x['instrumented_ms_per_step'])
show(d, d['config']['workload'])
for a in d['also']: show(a, a['workload'])
PY
