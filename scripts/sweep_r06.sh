#!/bin/bash
# round 6: switch points of the schedule re-swept with the merged bulk launches (same box, alternating rounds)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
N=${1:-16384}
run() { echo -n "$* : "; env "$@" python3 scripts/time_fit.py $N 2>&1 | grep -v amdgpu; }
for r in 1 2; do
  run X=0
  run AGP_MERGE_ABOVE=0
  run AGP_FP64_NBO=768
  run AGP_FP64_NBO=1024
  run AGP_FP64_NBO=1024 AGP_X_NBO_WIDE_ABOVE=6144
  run AGP_X_FUSED_BELOW=16384
  run AGP_X_FUSED_BELOW=8192
  run AGP_X_INNER_LEFT_ABOVE=16384
  run AGP_X_INNER_LEFT_ABOVE=0
  run AGP_X_MASK_BELOW=0
  run AGP_X_MASK_BELOW=6656
  run AGP_X_THROTTLE_BELOW=0
  run AGP_X_THROTTLE_BELOW=6144
  run AGP_STEP_BELOW=5632
  run AGP_STEP_BELOW=3584
done
