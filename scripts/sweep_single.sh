#!/bin/bash
# fits/s at N = 16384 (and ms per fit at N = 4096) against the remaining size below which the factorisation stays on one stream
for sb in ${SB:-0 1024 1536 2048 3072}; do
  echo -n "AGP_SINGLE_BELOW=$sb AGP_FUSED_BELOW=${FB:-4608}: "
  AGP_FUSED_BELOW=${FB:-4608} AGP_SINGLE_BELOW=$sb python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-predict --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), 'fits/s', round(d['ms_per_step'],2), 'ms')"
  AGP_FUSED_BELOW=${FB:-4608} AGP_SINGLE_BELOW=$sb python scripts/time_config2.py 2048 4096 2>&1 | cut -c1-60
done
