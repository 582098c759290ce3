set -o pipefail
R=$PWD; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 $R/bench.py --no-cpu-baseline --no-also --labels 1 --length 5000 --dtype bf16 --steps 10 --warmup 3 --priming 0 > $O/prof_c5.json 2> $O/prof_c5.err || exit 1
python3 $R/tools/prof_summary.py $O/prof_c5 18 40 > $O/prof_c5_summary.txt
cat $O/prof_c5_summary.txt
