cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mixed_precision_gpu.py tests/test_full_size_configs_gpu.py -x -q -m gpu 2>&1 | tail -8
python3 scripts/time_mixed.py 32768 2>&1 | grep -v amdgpu.ids | tail -4
AGP_SWEEP_COOP=0 python3 scripts/time_mixed.py 32768 2>&1 | grep -v amdgpu.ids | tail -4
