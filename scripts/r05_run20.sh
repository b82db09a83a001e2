cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=6 2>&1 | tail -14 ) 2>&1 | tail -18
python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 300 gpurun_out/bench_final.json; tail -3 gpurun_out/bench_final.err
python3 scripts/time_sharded_rank.py 16384 2>&1 | grep -v amdgpu.ids | tail -7
