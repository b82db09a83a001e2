#!/bin/bash
# Sweep the outer-block switch points (remaining size above which NBO = 512 / 256 is used); SWEEP="a,b c,d ..." overrides.
for sw in ${SWEEP:-"2048,1024" "0,0" "1024,512" "3072,1024" "4096,2048" "5120,2560" "6144,2816" "6144,1024" "4096,1024" "8192,4096"}; do
  AGP_NBO_SWITCH=$sw python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-predict 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('AGP_NBO_SWITCH=$sw', round(d['value'],3), 'fits/s', round(d['ms_per_step'],3), 'ms, factor', round(d['stages_ms_per_fit']['factor'],3))"
done
