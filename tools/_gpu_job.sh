python -m pytest tests/test_rccl_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
