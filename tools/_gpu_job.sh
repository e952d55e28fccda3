mkdir -p gpurun_out/r2x
python -m pytest tests/test_gemm_fp8.py tests/test_vit_fp8.py tests/test_siglip.py tests/test_rowops.py tests/test_convnext.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
python tools/bench_cfg5.py 128 3 | tee gpurun_out/r2x/cfg5.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k:round(v,1) for k,v in d.items() if 'images_per_sec' in k or 'ms_per_step' in k})"
