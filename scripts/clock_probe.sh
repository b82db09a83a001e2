#!/bin/bash
# The XCD-aware tile order of the bulk update against the default one: fits/s, SCLK held by the kernel (probe build),
# and HBM/MALL traffic per launch (separate --pmc passes).  On the GPU box:  bash scripts/clock_probe.sh r03
set -u
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/prof_$R"
mkdir -p "$OUT"
cd "$ROOT"
for m in 0 2; do
  echo -n "bench AGP_XCD_REMAP=$m: "
  AGP_XCD_REMAP=$m python3 bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-predict --no-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), 'fits/s', round(d['ms_per_step'],2), 'ms, bulk kernel', round(d['roofline']['achieved'],2), 'TFLOP/s')"
done
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  AGP_XCD_REMAP=2 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/xcd2_pmc_$c" -o bench -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-predict --no-configs > /dev/null 2> "$OUT/xcd2_pmc_$c.err"
done
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
for c, mult in (("FETCH_SIZE", 2048.), ("WRITE_SIZE", 1024.)):
    per = {}
    for f in glob.glob(os.path.join(out, f"xcd2_pmc_{c}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("agp::trailing_update_kernel") and r["Counter_Name"] == c:
                per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.) + float(r["Counter_Value"])
    if per:
        print(f"AGP_XCD_REMAP=2 trailing_update_kernel {c}: {len(per)} launches, {sum(per.values()) / len(per) * mult / 1e9:.3f} GB per launch (x{mult:.0f} B per count)")
PY
# SCLK: probe build of both libraries (in place on the box's scratch copy)
cd "$ROOT/albatross_amd/csrc" && touch gemm.hip && make -s -j16 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-result -DAGP_CLOCK_PROBE" > /dev/null 2>&1
cd "$ROOT"
for m in 0 2; do AGP_XCD_REMAP=$m python3 scripts/clock_probe.py 2>/dev/null; done
