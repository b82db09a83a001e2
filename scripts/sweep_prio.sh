#!/bin/bash
# fits/s at N = 16384 for the four combinations of wave priority (panel-chain kernels 3 / 0, bulk update 0 / 3): rebuilds the
# library in place for every combination (run on the GPU box's scratch copy), then restores the default build
cd "$(dirname "$0")/.."
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-result"
for c in 3 0; do for b in 0 3; do
  touch albatross_amd/csrc/*.hip
  make -s -j16 -C albatross_amd/csrc HIPFLAGS="$BASE -DAGP_CHAIN_PRIO=$c -DAGP_BULK_PRIO=$b" > /dev/null 2>&1
  echo -n "chain prio $c, bulk prio $b: "
  python3 bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-predict --no-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), 'fits/s', round(d['ms_per_step'],2), 'ms, bulk kernel', round(d['roofline']['achieved'],2), 'TFLOP/s')"
done; done
touch albatross_amd/csrc/*.hip; make -s -j16 -C albatross_amd/csrc > /dev/null 2>&1
