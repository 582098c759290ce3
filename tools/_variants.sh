set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -q > $O/r02_final_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r02_final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py --steps 20 --warmup 5 > $O/r02_bench_timed.json 2>/dev/null ) 2>&1 | grep real
bash tools/collect_profiles.sh r02 > $O/r02_collect.log 2>&1; echo "collect rc=$?"; tail -3 $O/r02_collect.log
