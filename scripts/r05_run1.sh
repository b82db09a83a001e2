cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gp_gpu.py tests/test_fit_batch_gpu.py tests/test_robustness_gpu.py tests/test_device_inputs_gpu.py -x -q -m gpu 2>&1 | tail -15
bash scripts/r05_baseline.sh coop1 2>&1 | tail -12
AGP_BACKSUB_COOP=0 python3 scripts/trace_config2_api.py
TRACE_N=16384 python3 scripts/trace_config2_api.py
TRACE_N=16384 AGP_BACKSUB_COOP=0 python3 scripts/trace_config2_api.py
