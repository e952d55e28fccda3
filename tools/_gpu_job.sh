mkdir -p gpurun_out/r2x
python -m pytest tests/test_convnext.py tests/test_parity_fullsize_gpu.py tests/test_gemm.py tests/test_conv.py -m gpu -x -q 2>&1 | tail -3
echo "n256 default(128)"; python tools/bench_cfg3.py 512 4 | tee gpurun_out/r2x/cfg3.json | cut -c1-330
echo "n256 min 256"; VDK_GEMM_MIN_N256=256 python tools/bench_cfg3.py 512 4 | cut -c1-330
