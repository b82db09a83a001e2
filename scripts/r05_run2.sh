cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_fit_batch_gpu.py tests/test_gp_gpu.py -x -q -m gpu 2>&1 | tail -15
FIT_BATCHES=1,8,32,64,256 python3 scripts/time_fit_batch.py 512 1024 2>&1 | grep -v amdgpu.ids
FIT_BATCHES=8,64,256 python3 scripts/time_fit_batch.py 2048 2>&1 | grep -v amdgpu.ids
AGP_FIT_SOLO=0 FIT_BATCHES=8,32,64,256 python3 scripts/time_fit_batch.py 512 1024 2>&1 | grep -v amdgpu.ids
for n in 512 1024; do TRACE_N=$n python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu.ids; done
