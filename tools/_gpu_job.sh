mkdir -p gpurun_out/r2h
python tools/cbir_capsweep.py 2>/dev/null | tee gpurun_out/r2h/capsweep.json
VDK_CBIR_PIPELINE=1 python -m pytest tests/test_cbir.py -m gpu -q 2>&1 | tail -2
