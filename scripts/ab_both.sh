#!/bin/bash
# Same-box A/B of BOTH libraries (product + debug) against the variants under scripts/variants/<name>/ (scripts/build_variant.sh):
#   bash scripts/ab_both.sh "<command>" [rounds]
# like scripts/ab.sh, for measurements that go through libalbatross_amd_debug.so.  Restores both libraries on every exit path.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
CMD=${1:?usage: ab_both.sh "<command>" [rounds]}
ROUNDS=${2:-2}
mkdir -p /tmp/ab_both_current && cp albatross_amd/libalbatross_amd.so albatross_amd/libalbatross_amd_debug.so /tmp/ab_both_current/ || exit 1
trap 'cp /tmp/ab_both_current/*.so "$ROOT/albatross_amd/"' EXIT
for r in $(seq "$ROUNDS"); do
  for v in current $(ls -d scripts/variants/*/ 2>/dev/null | xargs -n1 basename); do
    if [ "$v" = current ]; then cp /tmp/ab_both_current/*.so albatross_amd/; else cp "scripts/variants/$v/"*.so albatross_amd/; fi
    echo "[$v]"
    $CMD 2>&1 | grep -v amdgpu.ids
  done
done
