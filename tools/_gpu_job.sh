python -m pytest tests/test_parity_fullsize_gpu.py tests/test_convnext.py tests/test_fullsize_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -3
python tools/bench_cfg3.py 512 4 1000000 1 2>&1 | tail -1 | cut -c1-400
