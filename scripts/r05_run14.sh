cd $GRAFT_REPO_ROOT
cp albatross_amd/libalbatross_amd.so /tmp/lib_current.so
for v in current old8698727 current old8698727; do
  if [ "$v" = current ]; then cp /tmp/lib_current.so albatross_amd/libalbatross_amd.so; else cp scripts/variants/lib_$v.so albatross_amd/libalbatross_amd.so; fi
  echo -n "$v: "; python3 bench.py --no-cpu-baseline --no-configs --no-predict 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
cp /tmp/lib_current.so albatross_amd/libalbatross_amd.so
