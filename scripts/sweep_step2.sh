#!/bin/bash
# step launches: placeholder workgroups next to the critical ones (AGP_STEP_HOLD: 0 none, 1 adaptive (default), 10 the factoring
# workgroup and the nine that feed it, 1000 all critical ones always), chol.hip: panel_phase
for hold in ${HOLDS:-0 1 10 1000}; do
  echo -n "AGP_STEP_HOLD=$hold: "
  AGP_STEP_HOLD=$hold python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-predict --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), 'fits/s', round(d['ms_per_step'],2), 'ms')"
  AGP_STEP_HOLD=$hold python scripts/time_config2.py 1024 2048 4096 8192 2>&1 | cut -c1-60
done
