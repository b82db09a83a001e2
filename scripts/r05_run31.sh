cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_mixed_precision_gpu.py tests/test_gp_gpu.py -m gpu -x -q 2>&1 | tail -n 4
for pad in 0 8192 0 8192; do echo "AGP_BF16X3_LDS_PAD=$pad"; AGP_BF16X3_LDS_PAD=$pad python3 scripts/time_mixed.py 2>&1 | grep "N=32768"; done
python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu; TRACE_N=16384 python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu
python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['configs']['config2']['fit_ms'], d['configs']['config4']['mixed_fit_ms'])"
