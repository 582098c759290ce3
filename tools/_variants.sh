set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -q -x > $O/r02m_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r02m_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --labels 1 --length 5000 --dtype bf16 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms']); [print(r['entry'], r['avg_us'], r['frac']) for r in d['layers']]; print(d['instrumented_ms_per_step'])"
