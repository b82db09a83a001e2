#!/bin/bash
# fits/s at N = 16384: CUs kept by the end-phase bulk stream x remaining size at which the bulk updates move to it
for cus in ${CUS:-0 192 208 224 240}; do for below in ${BELOW:-4608 6656 8704}; do
  echo -n "AGP_MASK_CUS=$cus AGP_MASK_BELOW=$below THROTTLE=${THR:-8192}: "
  AGP_THROTTLE_BELOW=${THR:-8192} AGP_MASK_CUS=$cus AGP_MASK_BELOW=$below python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-predict 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), 'fits/s', round(d['ms_per_step'],2), 'ms', {k: round(v,2) for k,v in d['stages_ms_per_fit'].items()})"
  [ $cus = 0 ] && break
done; done
