cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gp_gpu.py tests/test_fit_batch_gpu.py tests/test_robustness_gpu.py -m gpu -x -q 2>&1 | tail -n 4
bash scripts/r05_ab.sh "512 1024 1280" 2
for m in 1280 1536 1792 2048; do
  for n in 1536 1792 2048; do echo -n "coop_max $m: "; AGP_BACKSUB_COOP_MAX=$m TRACE_N=$n python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu.ids; done
done
