#!/bin/bash
# Sweep the outer-block switch points (remaining size above which NBO = 512 / 256 is used).
for sw in "2048,1024" "0,0" "1024,512" "1024,0" "2048,0" "512,0" "3072,1024" "2048,512"; do
  AGP_NBO_SWITCH=$sw python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-predict | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$sw', round(d['value'],3), round(d['ms_per_step'],3), round(d['stages_ms_per_fit']['factor'],3))"
done
