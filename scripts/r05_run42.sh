cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gp_gpu.py tests/test_gram_gpu.py -m gpu -x -q 2>&1 | tail -n 3
python3 scripts/time_gram_trees.py 2>&1 | grep -v amdgpu
python3 scripts/time_predict_large.py 2>&1 | grep -v amdgpu | tail -n 12
