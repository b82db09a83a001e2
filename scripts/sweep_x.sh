B="python bench.py --no-cpu-baseline --no-predict --no-configs --steps 10 --warmup 2"
run() { echo "$1: $(env $1 $B 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3))')"; }
run "AGP_X_NONE=1"
for m in 10240 12288 16384; do run "AGP_X_MASK=$m"; done
for m in 10240 12288; do run "AGP_X_MASK=$m AGP_X_THROTTLE=$m"; done
for k in 232 240 248; do run "AGP_X_KEEP=$k"; run "AGP_X_KEEP=$k AGP_X_MASK=12288"; done
for f in 6144 8192; do run "AGP_X_FUSED=$f"; done
for f in 4096 8192 100000; do run "AGP_X_INNER=$f"; done
run "AGP_X_NONE=1"
