cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_mixed_precision_gpu.py tests/test_gp_gpu.py -x -q -m gpu 2>&1 | tail -5
python3 scripts/time_bf16x3.py 15872 30720 2>&1 | grep -v amdgpu.ids
for n in 512 1024 2048 4096 16384; do TRACE_N=$n python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu.ids; done
echo "coop backsub everywhere"
for n in 2048 4096 16384; do AGP_BACKSUB_COOP_MAX=100000 TRACE_N=$n python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu.ids; done
python3 scripts/time_mixed.py 32768 2>&1 | grep -v amdgpu.ids | tail -2
AGP_SWEEP_COOP=0 python3 scripts/time_mixed.py 32768 2>&1 | grep -v amdgpu.ids | tail -2
