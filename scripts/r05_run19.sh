cd $GRAFT_REPO_ROOT
for i in 1 2 3; do BENCH_DEBUG_STEPS=1 python3 bench.py --no-cpu-baseline --no-configs --no-predict 2>&1 | grep "bench.py rank\|value" | cut -c1-200; done
