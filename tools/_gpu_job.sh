mkdir -p gpurun_out/r2m
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_parity_bf16.py tests/test_vit.py -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o vit -- python $R/bench.py --steps 10 --warmup 2 --no-parity --no-cbir --no-cpu-baseline > /tmp/prof_stdout.txt 2>&1
cd $R
db=$(find /tmp/prof -name "*.db" | head -1)
python tools/rocpd_stats.py "$db" > gpurun_out/r2m/kernel_stats.txt
head -24 gpurun_out/r2m/kernel_stats.txt | cut -c1-140
python bench.py --steps 20 --warmup 3 --no-parity --no-cbir --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r2m/bench.json
python -c "
import json;d=json.load(open('gpurun_out/r2m/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_ms'])"
