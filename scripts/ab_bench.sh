#!/bin/bash
# A/B of one environment switch on the headline bench: scripts/ab_bench.sh VAR a b [reps]
var=$1; a=$2; b=$3; reps=${4:-2}
for r in $(seq $reps); do
  for v in $a $b; do
    env $var=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predict 2>/dev/null | tail -1 > /tmp/ab_line.json
    python - "$var" "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
r = d["roofline"]
print(sys.argv[1], sys.argv[2], "fits/s", round(d["value"], 3), "ms", round(d["ms_per_step"], 3), "roofline", r["achieved"], r["frac"])
PY
  done
done
