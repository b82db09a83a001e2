#!/bin/bash
# round-5 reference numbers of the chain-bound regime on ONE box (before / after the persistent fit kernel)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-before}
OUT="$ROOT/gpurun_out/r05_$TAG"
mkdir -p "$OUT"
cd "$ROOT"
FIT_BATCHES=1,32,256 python3 scripts/time_fit_batch.py 512 1024 > "$OUT/time_fit_batch.txt" 2>&1
FIT_BATCHES=1,8 python3 scripts/time_fit_batch.py 2048 4096 >> "$OUT/time_fit_batch.txt" 2>&1
for n in 512 1024 2048 4096; do TRACE_N=$n python3 scripts/trace_config2_api.py >> "$OUT/fit_resident.txt" 2>&1; done
cd /tmp && export TMPDIR=/tmp
for n in 512 4096; do
  TRACE_N=$n rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_n$n" -o t -- python3 "$ROOT/scripts/trace_config2_api.py" > "$OUT/trace_n$n.log" 2>&1
done
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
cd "$ROOT"
tail -n 20 "$OUT/time_fit_batch.txt" "$OUT/fit_resident.txt"
