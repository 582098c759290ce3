set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "bf16" > $O/r02n_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r02n_tests.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --labels 1 --length 5000 --dtype bf16 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms']['median']); print(' '.join(f\"{r['op'][0]}{r['c_in']}={r['avg_us']}\" for r in d['layers'])); print(d['instrumented_ms_per_step'])"
done
