mkdir -p gpurun_out/r2j
python -m pytest tests/test_parity_fullsize_gpu.py -m gpu -q -s -k resnet 2>&1 | tail -30 | cut -c1-1500 | tee gpurun_out/r2j/parity_fullsize.txt
