cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_distributed_gpu.py tests/test_fit_batch_gpu.py tests/test_update_dense_gpu.py tests/test_cpp_host_gpu.py tests/test_robustness_gpu.py tests/test_linear_combination_gpu.py -m gpu -x -q 2>&1 | tail -n 8
