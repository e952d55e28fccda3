mkdir -p gpurun_out/r2n
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_heads.py -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -3
python tools/bench_head.py | tee gpurun_out/r2n/head.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o head -- python $R/tools/bench_head.py > /tmp/prof_stdout.txt 2>&1
cd $R
db=$(find /tmp/prof -name "*.db" | head -1)
python tools/rocpd_stats.py "$db" > gpurun_out/r2n/head_kernel_stats.txt
head -24 gpurun_out/r2n/head_kernel_stats.txt | cut -c1-150
