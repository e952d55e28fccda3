mkdir -p gpurun_out/r2i
python tools/bench_gemm_fp8.py 2>&1 | tail -1 | tee gpurun_out/r2i/gemm_fp8.json
