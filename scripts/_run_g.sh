mkdir -p gpurun_out/r3m
(timeout 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3m/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3m/gpu_tests.log); tail -3 gpurun_out/r3m/gpu_tests.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3m/bench.json 2>gpurun_out/r3m/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r3m/bench.json')); print(d['value'], d['ms_per_step'], d['predict']); print({k:(v.get('fit_ms') or v.get('mixed_fit_ms')) for k,v in d['configs'].items()})"
