set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "bf16" > $O/r02h_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r02h_tests.log
python tools/stamp_fwd.py --bf16 --long 2>&1 | tail -4 | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['block'], 'wgs', d['workgroups'], 'span', d['kernel_span_us'], 'wg_med', d['wg_total_median_us'], d['median_us'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --labels 1 --length 5000 --dtype bf16 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms']); [print(r['entry'], r['avg_us'], r['frac']) for r in d['layers']]; print(d['instrumented_ms_per_step'])"
