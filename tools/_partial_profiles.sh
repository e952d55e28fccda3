R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BARGS="--steps 20 --warmup 3 --no-parity --no-cpu-baseline --no-cbir --no-cfg5 --no-swin"
rocprofv3 --kernel-trace --stats -d /tmp/p_trace -o t -- python $R/bench.py $BARGS > $O/trace_stdout.txt 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/p_trace -name "*.db" | head -1) > $O/bench_kernel_stats.txt
cd $R
T0=$(date +%s); python bench.py > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "default bench.py wall: $(( $(date +%s) - T0 )) s" | tee $O/bench_wall.txt
tail -1 $O/bench_stdout.txt > $O/bench.json
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['traffic'],d['roofline']['algorithmic_bytes_per_launch']);print(d['swin']['ms_per_step'],d['cbir']['ms_per_search'],d['cfg5']['sam_bf16']['images_per_sec'],d['cfg5']['sam_fp8']['images_per_sec'])"
