#!/bin/bash
# fits/s at N = 16384: CUs kept by the end-phase bulk stream x remaining size at which it takes over x remaining size at
# which the fused panel kernel takes over
for cus in ${CUS:-208 224 240}; do for mb in ${MB:-8704 6656}; do for fb in ${FB:-4608 8704}; do
  echo -n "AGP_MASK_CUS=$cus AGP_MASK_BELOW=$mb AGP_FUSED_BELOW=$fb: "
  AGP_MASK_CUS=$cus AGP_MASK_BELOW=$mb AGP_FUSED_BELOW=$fb python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-predict --no-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), 'fits/s', round(d['ms_per_step'],2), 'ms')"
done; done; done
