mkdir -p gpurun_out/r2o
python -m pytest tests/test_siglip.py tests/test_vit.py tests/test_parity_bf16.py tests/test_heads.py tests/test_gemm.py -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -4
timeout 600 python tools/bench_cfg5.py 128 3 2>&1 | tail -2 | tee gpurun_out/r2o/cfg5.json
