# same-box A/B of library variants (scripts/variants/lib_<name>.so against the in-tree build): usage r05_ab.sh "<sizes>" [rounds]
cd $GRAFT_REPO_ROOT
SIZES=${1:-"512 4096"}
ROUNDS=${2:-2}
cp albatross_amd/libalbatross_amd.so /tmp/lib_current.so
for r in $(seq $ROUNDS); do
  for v in current $(ls scripts/variants | sed 's/^lib_//; s/\.so$//'); do
    if [ "$v" = current ]; then cp /tmp/lib_current.so albatross_amd/libalbatross_amd.so; else cp scripts/variants/lib_$v.so albatross_amd/libalbatross_amd.so; fi
    for n in $SIZES; do echo -n "$v "; TRACE_N=$n python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu.ids; done
  done
done
cp /tmp/lib_current.so albatross_amd/libalbatross_amd.so
