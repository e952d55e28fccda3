mkdir -p gpurun_out/r2x
python -m pytest tests/test_attention.py tests/test_rowops.py tests/test_siglip.py -m gpu -x -q 2>&1 | tail -3
python tools/bench_attention.py 128 576 16 | tee gpurun_out/r2x/attn_576.json
python tools/bench_attention.py 256 257 16 | tee gpurun_out/r2x/attn_257.json
python tools/bench_cfg5.py 128 3 | tee gpurun_out/r2x/cfg5.json
