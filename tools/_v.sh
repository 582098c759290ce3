set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_abi.py -m gpu -x -q -k "bf16 or abi or n16" > gpurun_out/v_tests.log 2>&1
python bench.py --no-cpu-baseline --no-also --steps 30 --warmup 5 --dtype bf16 --length 5000 --labels 1 > gpurun_out/v_bench.log 2>&1
