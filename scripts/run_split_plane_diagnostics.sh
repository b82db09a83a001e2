#!/bin/bash
# The diagnostics behind DESIGN.md section 4 "Round 6" (on the GPU box, after scripts/build_variant.sh bf16_stamps -DAGP_BF16_STAMPS,
# f16_noload -DAGP_DIAG_F16_NOLOAD, f16_nostore -DAGP_DIAG_F16_NOSTORE): the split-plane bulk kernels alone, the fp16 x 2 kernel
# without its loads / LDS stores, the phase cycles of the bf16 x 3 kernel.  Restores the product libraries at the end.
cp albatross_amd/libalbatross_amd.so /tmp/a.so; cp albatross_amd/libalbatross_amd_debug.so /tmp/b.so
echo "== product build: scripts/time_bf16x3.py 30720 (5 launches each)"
python3 scripts/time_bf16x3.py 30720 2>&1 | grep "x [23]"
AGP_F16X2_TERMS=3 python3 scripts/time_bf16x3.py 30720 2>&1 | grep "fp16" | sed 's/$/  [AGP_F16X2_TERMS=3]/'
AGP_F16X2_CHUNK=64 python3 scripts/time_bf16x3.py 30720 2>&1 | grep "fp16" | sed 's/$/  [AGP_F16X2_CHUNK=64]/'
for v in f16_noload f16_nostore; do cp scripts/variants/$v/*.so albatross_amd/; echo "== diagnostic build $v (wrong results; see csrc/gemm_f16x2.hip)"; python3 scripts/time_bf16x3.py 30720 2>&1 | grep "fp16"; done
cp scripts/variants/bf16_stamps/*.so albatross_amd/
echo "== -DAGP_BF16_STAMPS build: scripts/probe_bf16x3.py (phase cycles of one workgroup of the bf16 x 3 kernel, held clock)"
python3 scripts/probe_bf16x3.py 15872 30720 2>&1 | grep -v amdgpu
cp /tmp/a.so albatross_amd/libalbatross_amd.so; cp /tmp/b.so albatross_amd/libalbatross_amd_debug.so
