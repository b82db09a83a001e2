cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_mixed_precision_gpu.py tests/test_gp_gpu.py -x -q -m gpu 2>&1 | tail -8
python3 scripts/time_bf16x3.py 15872 30720 2>&1 | grep -v amdgpu.ids
for n in 2048 3072 4096 4608; do TRACE_N=$n python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu.ids; done
python3 scripts/time_mixed.py 32768 2>&1 | grep -v amdgpu.ids | tail -12
AGP_MIXED_BF16=0 python3 scripts/time_mixed.py 32768 2>&1 | grep -v amdgpu.ids | tail -6
