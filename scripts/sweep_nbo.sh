#!/bin/bash
# experiment: outer-block switch points of the factorisation
for sw in "10240,6144" "99999,99999" "0,0" "12288,8192" "8192,4096" "6144,3072" "16384,8192" "4096,0"; do
  echo "AGP_NBO_SWITCH=$sw"; AGP_NBO_SWITCH=$sw python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"stages_ms_per_fit": {[^}]*}' | tr '\n' ' '; echo
done
