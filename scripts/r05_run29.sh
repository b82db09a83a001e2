cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_mixed_precision_gpu.py -m gpu -x -q 2>&1 | tail -n 4
for k in 2 1 2 1; do echo "AGP_BF16X3_KERNEL=$k"; AGP_BF16X3_KERNEL=$k python3 scripts/time_bf16x3.py 8192 15872 30720 2>&1 | grep "bf16 x 3"; done
for k in 2 1; do echo "AGP_BF16X3_KERNEL=$k"; AGP_BF16X3_KERNEL=$k python3 scripts/time_mixed.py 2>&1 | grep -v amdgpu.ids; done
