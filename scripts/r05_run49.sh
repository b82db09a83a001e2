cd $GRAFT_REPO_ROOT
time (timeout 1500 python3 -m pytest tests/ -m gpu -x -q 2>&1 | tail -n 4)
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
python3 bench.py > gpurun_out/bench_final4.json 2> gpurun_out/bench_final4.err; tail -c 300 gpurun_out/bench_final4.json
