#!/bin/bash
# fits/s at N = 16384 against the remaining size below which the fused panel kernel takes over (0 = always)
for fb in ${FB:-0 2048 4608 6656 8704 12288}; do
  echo -n "AGP_FUSED_BELOW=$fb: "
  AGP_FUSED_BELOW=$fb python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-predict --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), 'fits/s', round(d['ms_per_step'],2), 'ms', {k: round(v,2) for k,v in d['stages_ms_per_fit'].items()})"
done
echo -n "AGP_PANEL_FUSED=0: "
AGP_PANEL_FUSED=0 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-predict --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), 'fits/s', round(d['ms_per_step'],2), 'ms')"
