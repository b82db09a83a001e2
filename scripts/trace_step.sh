#!/bin/bash
# kernel timeline of one fit with the step tail: TRACE_N (default 4096), AGP_STEP_BELOW from the environment
cd /tmp && export TMPDIR=/tmp
N=${TRACE_N:-4096}
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_step_$N
rm -rf $OUT
TRACE_N=$N rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/trace_fit.py > /dev/null 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/trace_timeline.py $f full > $GRAFT_REPO_ROOT/gpurun_out/timeline_step_n$N.txt
rm -rf $OUT
