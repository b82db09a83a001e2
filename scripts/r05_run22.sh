cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gp_gpu.py tests/test_fit_batch_gpu.py tests/test_robustness_gpu.py -m gpu -x -q 2>&1 | tail -n 4
bash scripts/r05_ab.sh "512 1024 1280 2048 4096" 2
echo "coop up to 2048"
AGP_BACKSUB_COOP_MAX=2048 TRACE_N=2048 python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu.ids
AGP_BACKSUB_COOP_MAX=1536 TRACE_N=1536 python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu.ids
TRACE_N=1536 python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu.ids
