cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fit_batch_gpu.py tests/test_gp_gpu.py tests/test_robustness_gpu.py -x -q -m gpu 2>&1 | tail -15
for n in 512 1024 2048 4096 16384; do TRACE_N=$n python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu.ids; done
python3 scripts/time_panel.py 2>&1 | grep -v amdgpu.ids | tail -20
