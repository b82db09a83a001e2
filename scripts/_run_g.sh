mkdir -p gpurun_out/r3j
python scripts/time_sharded_rank.py 16384 > gpurun_out/r3j/time_sharded_rank.txt 2>&1; tail -5 gpurun_out/r3j/time_sharded_rank.txt
(timeout 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3j/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3j/gpu_tests.log); tail -3 gpurun_out/r3j/gpu_tests.log
