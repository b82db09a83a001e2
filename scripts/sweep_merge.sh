#!/bin/bash
# fits/s at N = 16384 against the remaining size above which U1 is merged into the bulk update (0 = never)
for ma in ${MA:-0 8704 10240 12288 6656}; do
  echo -n "AGP_MERGE_ABOVE=$ma: "
  AGP_MERGE_ABOVE=$ma python3 bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-predict --no-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), 'fits/s', round(d['ms_per_step'],2), 'ms, bulk kernel', round(d['roofline']['achieved'],2), 'TFLOP/s,', d['roofline']['launches_per_fit'], 'launches, self-check', d['self_check']['max_rel_residual'])"
done
