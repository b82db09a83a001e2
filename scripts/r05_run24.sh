cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gp_gpu.py tests/test_fit_batch_gpu.py tests/test_robustness_gpu.py tests/test_device_inputs_gpu.py -m gpu -x -q 2>&1 | tail -n 4
cp albatross_amd/libalbatross_amd.so /tmp/lib_current.so
for r in 1 2; do
for v in current head; do
  if [ "$v" = current ]; then cp /tmp/lib_current.so albatross_amd/libalbatross_amd.so; else cp scripts/variants/lib_$v.so albatross_amd/libalbatross_amd.so; fi
  echo "== $v"
  FIT_BATCHES=8,32,256 python3 scripts/time_fit_batch.py 512 1024 2>&1 | grep -v amdgpu.ids | cut -c1-120
done
done
cp /tmp/lib_current.so albatross_amd/libalbatross_amd.so
