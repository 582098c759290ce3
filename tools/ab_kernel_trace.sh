#!/bin/bash
# kernel-trace of a short headline bench per library variant (tools/_build/libecg_hip_<v>.so, "cur" = in-tree): prints the tail kernels
O=$1; shift
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = cur ]; then unset ECG_HIP_LIB; else export ECG_HIP_LIB=$R/tools/_build/libecg_hip_$v.so; fi
  rm -rf $O/prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -- python3 $R/bench.py --no-cpu-baseline --no-also --priming-seconds 0 --priming 0 --steps 20 --warmup 5 --model ${MODEL:-cnn} > $R/$O/bench_$v.json 2> $R/$O/bench_$v.err || { tail -3 $R/$O/bench_$v.err; exit 1; }
  echo "== $v"; python3 $R/tools/prof_summary.py $R/$O/prof_$v 50 40 | grep -i "${PATTERN:-tail\|bce\|linear_wgrad\|kernel time}"
done
