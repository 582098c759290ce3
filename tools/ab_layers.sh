#!/bin/bash
# same-box A/B of several libraries with tools/layer_bench.py: interleaved runs.
#   VARIANTS="prev cur cs2" bash tools/ab_layers.sh <out dir> <layer_bench args...>     ("cur" = the in-tree library)
O=$1; shift
V=${VARIANTS:-prev cur}
mkdir -p $O
for rep in 1 2 3; do
  for v in $V; do
    if [ $v = cur ]; then unset ECG_HIP_LIB; else export ECG_HIP_LIB=$PWD/tools/_build/libecg_hip_$v.so; fi
    python tools/layer_bench.py "$@" --tag $v > $O/${v}_$rep.txt 2>&1 || { tail -3 $O/${v}_$rep.txt; exit 1; }
  done
done
python - "$O" $V <<'PY'
import json, sys, glob, collections
O, V = sys.argv[1], sys.argv[2:]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for v in V:
    for f in sorted(glob.glob(f"{O}/{v}_[0-9].txt")):
        for l in open(f):
            if l.startswith("{") and '"op"' in l:
                d = json.loads(l); res[(d["block"], d["op"])][v].append(d["us"])
tot = collections.defaultdict(float)
for k in sorted(res):
    row = f"block {k[0]} {k[1]:6s}"
    for v in V:
        xs = sorted(res[k][v]); med = xs[len(xs) // 2]; tot[v] += med
        row += f"  {v} {med:7.1f}"
    print(row)
print("sum         " + "".join(f"  {v} {tot[v]:7.1f}" for v in V))
PY
