cd $GRAFT_REPO_ROOT
BENCH_DEBUG_STEPS=1 python3 bench.py --no-cpu-baseline --no-configs --no-predict 2>&1 | grep "bench.py rank"
