set -o pipefail
O=gpurun_out
for v in "ECG_HIP_PRIO=0" "ECG_HIP_PRIO=1" "ECG_HIP_PRIO=2"; do
  echo "== $v"
  env $v python tools/stamp_fwd.py 2>&1 | tail -4 | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['block'], 'span', d['kernel_span_us'], 'end_spread', d['end_spread_us'], 'wg_med', d['wg_total_median_us'], d['median_us'])"
  env $v python tools/layer_bench.py --tag "$v" 2>/dev/null | grep -v wgrad | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print(' '.join(f\"b{r['block']}{r['op'][0]}={r['us']}\" for r in rows if 'op' in r), 'sum', round(sum(r['us'] for r in rows if 'op' in r),1))"
done
