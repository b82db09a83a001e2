#!/bin/bash
# width of the explicit diagonal-block inverses of the backward substitution (AGP_WIDE_BACKSOLVE; 0 = 128-row steps)
for w in ${@:-0 256 512 1024 2048}; do
  AGP_WIDE_BACKSOLVE=$w python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-predict 2>/dev/null | tail -1 > /tmp/bs.json
  python - $w <<'PY'
import json, sys
d = json.load(open("/tmp/bs.json"))
print("width", sys.argv[1], "fits/s", round(d["value"], 3), "ms", round(d["ms_per_step"], 3), "backward solve ms", round(d["stages_ms_per_fit"]["backward_solve"], 3))
PY
done
