mkdir -p gpurun_out/r2v
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_heads.py tests/test_face.py tests/test_parity_fullsize_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -3
python tools/bench_head.py | tee gpurun_out/r2v/head.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ph -o head -- python $R/tools/bench_head.py > /tmp/ph.log 2>&1
cd $R
python tools/rocpd_stats.py $(find /tmp/ph -name "*.db" | head -1) > gpurun_out/r2v/head_kernel_stats.txt
head -16 gpurun_out/r2v/head_kernel_stats.txt | cut -c1-140
