set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -q -k "full_size" > $O/r02i_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/r02i_tests.log
bash tools/collect_profiles.sh r02 > $O/r02_collect.log 2>&1; echo "collect rc=$?"; tail -5 $O/r02_collect.log
