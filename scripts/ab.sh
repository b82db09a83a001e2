#!/bin/bash
# Same-box A/B of library variants (the boxes of the pool differ by 7-9 % on these sizes): every
# scripts/variants/lib_<name>.so takes the place of the in-tree libalbatross_amd.so in turn, `rounds` alternating rounds.
#   bash scripts/ab.sh "<command>" [rounds]      e.g.  bash scripts/ab.sh "python3 scripts/time_fit.py 16384" 2
# The in-tree library is restored on EVERY exit path.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
CMD=${1:-"python3 scripts/time_fit.py 16384"}
ROUNDS=${2:-2}
LIB=albatross_amd/libalbatross_amd.so
cp "$LIB" /tmp/lib_current.so || exit 1
trap 'cp /tmp/lib_current.so "$ROOT/$LIB"' EXIT
for r in $(seq "$ROUNDS"); do
  for v in current $(ls scripts/variants 2>/dev/null | sed -n 's/^lib_\(.*\)\.so$/\1/p'); do
    if [ "$v" = current ]; then cp /tmp/lib_current.so "$LIB"; else cp "scripts/variants/lib_$v.so" "$LIB"; fi
    echo -n "[$v] "
    $CMD 2>&1 | grep -v amdgpu.ids
  done
done
