mkdir -p gpurun_out/r2k
python tools/bench_gemm_sk.py 65536 2>&1 | tail -1 | tee gpurun_out/r2k/gemm_sk_65536.json
