mkdir -p gpurun_out/r2x
python -m pytest tests/test_gemm_fp8.py tests/test_vit_fp8.py tests/test_rowops.py -m gpu -x -q 2>&1 | tail -3
python tools/bench_cfg5.py 128 3 | tee gpurun_out/r2x/cfg5.json
