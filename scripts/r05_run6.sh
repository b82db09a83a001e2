cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | tail -25 ) 2>&1
for sb in 4608 3584 3072 2560 2048; do echo "step_below $sb"; for n in 3072 4096 4608; do AGP_STEP_BELOW=$sb TRACE_N=$n python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu.ids; done; done
