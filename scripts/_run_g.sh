for tp in 1 2 4; do echo "AGP_PREDICT_TP=$tp"; AGP_PREDICT_TP=$tp python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['predict'])"; done
timeout 600 python -m pytest tests/test_gp_gpu.py tests/test_device_inputs_gpu.py -m gpu -x -q 2>&1 | tail -3
