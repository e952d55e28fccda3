mkdir -p gpurun_out/r2x
python -m pytest tests/test_attention.py -m gpu -x -q 2>&1 | tail -3
echo default; python tools/bench_attention.py 256 197 12 | tee gpurun_out/r2x/attn_197_default.json
echo streaming; VDK_ATTN_LONG_MIN=1 python tools/bench_attention.py 256 197 12 | tee gpurun_out/r2x/attn_197_streaming.json
python tools/bench_attention.py 128 576 16 | tee gpurun_out/r2x/attn_576.json
python tools/bench_attention.py 256 257 16 | tee gpurun_out/r2x/attn_257.json
python tools/bench_attention.py 128 577 16 | tee gpurun_out/r2x/attn_577.json
