cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_update_dense_gpu.py tests/test_linear_combination_gpu.py tests/test_fit_batch_gpu.py -x -q -m gpu 2>&1 | tail -6
bash scripts/r05_ab.sh "512 2048 4096 16384" 2
