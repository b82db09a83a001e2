#!/bin/bash
# fits/s at N = 16384 (and ms per fit at N = 2048 / 4096) against AGP_STEP_BELOW: the remaining size at or below which
# the factorisation runs one launch per panel on one stream (chol.hip: panel_phase step_mode)
for sb in ${SB:-0 1536 2048 3072 4096 4608}; do
  echo -n "AGP_STEP_BELOW=$sb: "
  AGP_STEP_BELOW=$sb python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-predict --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), 'fits/s', round(d['ms_per_step'],2), 'ms')"
  AGP_STEP_BELOW=$sb python scripts/time_config2.py 1024 2048 4096 8192 2>&1 | cut -c1-60
done
