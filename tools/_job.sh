cd /root/repo
timeout 3000 python -m pytest tests/ -x -q -m gpu > gpurun_out/gpu_tests_full.log 2>&1; grep -E "passed|failed" gpurun_out/gpu_tests_full.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
