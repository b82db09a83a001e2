#!/bin/bash
# All profile passes of a round, on the GPU box:  bash scripts/profile_round.sh r03
# Writes under gpurun_out/prof_<round>/ ; scripts/pmc_summary.py turns the CSVs into the
# summaries kept under profiles/<round>/.
set -u
R=${1:-r06}
# the repository root, resolved BEFORE the cd below (GRAFT_REPO_ROOT is only set on the gpurun box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/prof_$R"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
FAST="--no-cpu-baseline --no-predict --no-configs"
python3 "$B" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$B" --steps 5 --warmup 2 $FAST > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o bench -- python3 "$B" --steps 2 --warmup 1 $FAST > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o bench -- python3 "$B" --steps 2 --warmup 1 $FAST > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -o bench -- python3 "$B" --steps 2 --warmup 1 $FAST > /dev/null 2> "$OUT/pmc_mfma.err"
# keep the merge-back small: the raw traces are large
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
cd "$ROOT"
# the other measurements DESIGN.md quotes
python3 scripts/time_panel.py > "$OUT/time_panel_fused.txt" 2>&1
python3 scripts/time_gram_trees.py > "$OUT/time_gram_trees.txt" 2>&1
python3 scripts/time_mixed.py 32768 > "$OUT/time_mixed.txt" 2>&1
python3 scripts/time_sharded_rank.py 16384 65536 > "$OUT/time_sharded_rank.txt" 2>&1
AGP_SHARD_HOST_PACING=1 python3 scripts/time_sharded_rank.py 16384 > "$OUT/time_sharded_rank_hostpaced.txt" 2>&1
python3 scripts/time_sharded_rccl1.py > "$OUT/time_sharded_rccl1.txt" 2>&1
python3 scripts/time_config2.py 1024 2048 4096 8192 > "$OUT/time_config2.txt" 2>&1
python3 scripts/time_fit_batch.py > "$OUT/time_fit_batch.txt" 2>&1
python3 scripts/fit_vs_n.py > "$OUT/fit_vs_n.txt" 2>&1
python3 scripts/time_bf16x3.py 15872 30720 > "$OUT/time_bf16x3.txt" 2>&1
FIT_BATCHES=1,8,32,256 python3 scripts/time_fit_batch.py 512 1024 > "$OUT/time_fit_batch_256.txt" 2>&1
for n in 512 1024 2048 4096; do TRACE_N=$n python3 scripts/trace_config2_api.py; done > "$OUT/fit_resident.txt" 2>&1
ls -la "$OUT" "$OUT"/*/ | head -60
tail -1 "$OUT/bench_n1.json" | cut -c1-300
