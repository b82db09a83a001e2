cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests/test_distributed_gpu.py -x -q -m gpu 2>&1 | tail -5 ) 2>&1 | tail -8
python3 scripts/time_sharded_rank.py 16384 2>&1 | grep -v amdgpu.ids | tail -12
python3 bench.py > gpurun_out/bench_after_pool.json 2> gpurun_out/bench_after_pool.err; tail -c 600 gpurun_out/bench_after_pool.json
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_after_pool.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(d['configs']['config2']['fit_ms'])
for r in d['configs']['small_n_batched']['rows']:
    print(r['n'], r['batch'], round(r['ms_per_batch'],4), r.get('gpu_span_ms'))
print(d['cpu_baseline']['sample'])
PY
