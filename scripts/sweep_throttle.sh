#!/bin/bash
# fits/s at N = 16384: host-throttling threshold of the bulk-update launches (chol.hip: throttle_below) x tile shape of
# the small bulk updates (gemm.hip: AGP_TAIL_SPLIT 0 = 128 x 128 tiles only, 1 = 64 x 64 tails and small launches)
for split in 1 0; do for t in 0 8192; do
  echo -n "AGP_TAIL_SPLIT=$split AGP_THROTTLE_BELOW=$t: "
  AGP_TAIL_SPLIT=$split AGP_THROTTLE_BELOW=$t python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-predict 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), 'fits/s', round(d['ms_per_step'],2), 'ms', {k: round(v,2) for k,v in d['stages_ms_per_fit'].items()})"
done; done
