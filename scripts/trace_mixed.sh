cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_mixed -o t -- python3 scripts/trace_mixed.py > gpurun_out/trace_mixed.log 2>&1
f=$(find gpurun_out/trace_mixed -name "*kernel_trace.csv" | head -n 1)
python3 scripts/trace_timeline.py $f > gpurun_out/timeline_mixed.txt 2>&1
cp $(find gpurun_out/trace_mixed -name "*kernel_stats.csv" | head -n 1) gpurun_out/kernel_stats_mixed.csv
rm -rf gpurun_out/trace_mixed
grep "mixed fit" gpurun_out/trace_mixed.log
head -n 40 gpurun_out/timeline_mixed.txt
