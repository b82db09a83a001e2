mkdir -p gpurun_out/r3k
timeout 900 python -m pytest tests/test_gram_gpu.py tests/test_variant_gpu.py tests/test_linear_combination_gpu.py tests/test_mixed_precision_gpu.py -m gpu -x -q 2>&1 | tail -5
python scripts/time_gram_trees.py > gpurun_out/r3k/time_gram_trees.txt 2>&1; cat gpurun_out/r3k/time_gram_trees.txt
AGP_GRAM_PAIR2=0 python scripts/time_gram_trees.py 2>&1 | sed 's/^/PAIR2=0 /'
