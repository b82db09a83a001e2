#!/bin/bash
# kernel timeline of one rank's share of a sharded fit (AGP_SHARD_FAKE_WORLD, AGP_SHARD_HOST_PACING from the environment)
cd /tmp && export TMPDIR=/tmp
TAG=${1:-dev}
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_sharded_$TAG
rm -rf $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/${TRACE_SCRIPT:-trace_sharded.py} > /dev/null 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/trace_timeline.py $f full > $GRAFT_REPO_ROOT/gpurun_out/r04/timeline_sharded_$TAG.txt
rm -rf $OUT
