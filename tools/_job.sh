cd /root/repo
timeout 3000 python -m pytest tests/ -x -q -m gpu > gpurun_out/gpu_tests_full.log 2>&1; tail -5 gpurun_out/gpu_tests_full.log
