#!/bin/bash
# All profile passes of a round, on the GPU box:  bash scripts/profile_round.sh r01
# Writes under gpurun_out/prof_<round>/ ; scripts/pmc_summary.py turns the CSVs into the
# summaries kept under profiles/<round>/.
set -u
R=${1:-r02}
# the repository root, resolved BEFORE the cd below (GRAFT_REPO_ROOT is only set on the gpurun box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/prof_$R"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
python3 "$B" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$B" --steps 5 --warmup 2 --no-cpu-baseline --no-predict > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o bench -- python3 "$B" --steps 2 --warmup 1 --no-cpu-baseline --no-predict > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o bench -- python3 "$B" --steps 2 --warmup 1 --no-cpu-baseline --no-predict > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -o bench -- python3 "$B" --steps 2 --warmup 1 --no-cpu-baseline --no-predict > /dev/null 2> "$OUT/pmc_mfma.err"
# keep the merge-back small: the raw traces are large
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
ls -la "$OUT" "$OUT"/*/ | head -40
tail -1 "$OUT/bench_n1.json" | cut -c1-300
