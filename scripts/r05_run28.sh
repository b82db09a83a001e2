cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_bf
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCP|TCC|TA|TD)_[A-Z0-9_]+" | sort -u > gpurun_out/pmc_avail.txt
wc -l gpurun_out/pmc_avail.txt
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_bf/$tag -o t -- python3 scripts/time_bf16x3.py 30720 > gpurun_out/pmc_bf/$tag.log 2>&1
  f=$(find gpurun_out/pmc_bf/$tag -name "*counter_collection.csv" | head -n 1)
  echo "== $set"
  python3 - "$f" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-50:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if "trailing" in k or "update" in k:
        print(k, {c: f"{v:.4g}" for c, v in d.items()})
P
done
rm -rf gpurun_out/pmc_bf/*/ 
