# SQ / TCC counters of the three bulk update kernels (fp64, fp32, bf16 x 3) in isolation, one --pmc set per pass:
#   bash scripts/pmc_bulk_kernels.sh [M]      -> stdout (profiles/r05/pmc_bulk_kernels_*.txt)
# (FETCH_SIZE and WRITE_SIZE need a pass EACH - scripts/profile_round.sh -: asked for together, rocprofv3 hung until the
# time limit on this image)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
M=${1:-30720}
mkdir -p gpurun_out/pmc_bf
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_bf/$tag -o t -- python3 scripts/time_bf16x3.py $M > gpurun_out/pmc_bf/$tag.log 2>&1
  f=$(find gpurun_out/pmc_bf/$tag -name "*counter_collection.csv" | head -n 1)
  echo "== $set"
  python3 - "$f" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-50:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if "trailing" in k or "update" in k:
        print(k, {c: f"{v:.4g}" for c, v in d.items()})
P
done
rm -rf gpurun_out/pmc_bf
