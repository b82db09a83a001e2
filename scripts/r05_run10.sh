cd $GRAFT_REPO_ROOT
cp albatross_amd/libalbatross_amd_debug.so /tmp/dbg_current.so
for v in current bf10 bf01 bf11; do
  if [ "$v" = current ]; then cp /tmp/dbg_current.so albatross_amd/libalbatross_amd_debug.so; else cp scripts/variants/libdbg_$v.so albatross_amd/libalbatross_amd_debug.so; fi
  echo "variant $v (pin agpr)"; python3 scripts/time_bf16x3.py 15872 30720 2>&1 | grep "bf16"
done
cp /tmp/dbg_current.so albatross_amd/libalbatross_amd_debug.so
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_mixed_precision_gpu.py tests/test_gp_gpu.py -x -q -m gpu 2>&1 | tail -3
python3 scripts/time_mixed.py 32768 2>&1 | grep -v amdgpu.ids | tail -2
