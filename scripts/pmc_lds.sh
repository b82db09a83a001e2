# LDS bank conflicts / waits per kernel of a fit: bash scripts/pmc_lds.sh <N>   (writes gpurun_out/pmc_lds_n<N>.txt)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
N=${1:-4096}
mkdir -p gpurun_out/pmc_lds
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $set | cut -c1-20 | tr ' ' '_')
  TRACE_N=$N rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_lds/$tag -o t -- python3 scripts/trace_config2_api.py > gpurun_out/pmc_lds/$tag.log 2>&1
  f=$(find gpurun_out/pmc_lds/$tag -name "*counter_collection.csv" | head -n 1)
  echo "== N=$N: $set"
  python3 - "$f" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-48:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
    print(f"{k:50s}", {c: f"{v:.3g}" for c, v in d.items()})
P
done
rm -rf gpurun_out/pmc_lds
