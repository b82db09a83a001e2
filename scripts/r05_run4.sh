cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fit_batch_gpu.py tests/test_gp_gpu.py tests/test_robustness_gpu.py tests/test_update_dense_gpu.py -x -q -m gpu 2>&1 | tail -8
for n in 512 1024 2048 4096 16384; do TRACE_N=$n python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu.ids; done
for b in 0 300 500 800 1200 2000; do echo "budget $b"; for n in 3072 4096 4608; do AGP_STEP_BUDGET=$b TRACE_N=$n python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu.ids; done; done
for b in 0 500 800 1200; do echo "budget $b"; AGP_STEP_BUDGET=$b TRACE_N=16384 python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu.ids; done
python3 scripts/time_panel.py 2>&1 | grep -v amdgpu.ids | tail -8
