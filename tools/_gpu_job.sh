mkdir -p gpurun_out/r2x
python -m pytest tests/ -m gpu -x -q > gpurun_out/r2x/gpu_tests_full.txt 2>&1; echo "rc=$?"
grep -E "passed|failed|error" gpurun_out/r2x/gpu_tests_full.txt | tail -5
