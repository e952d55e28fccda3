R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BARGS="--steps 20 --warmup 3 --no-parity --no-cpu-baseline --no-cbir --no-cfg5 --no-swin"
rocprofv3 --kernel-trace --stats -d /tmp/p_trace -o t -- python $R/bench.py $BARGS > $O/trace_stdout.txt 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/p_trace -name "*.db" | head -1) > $O/bench_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/p_$c -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-parity --no-cpu-baseline --no-cbir --no-cfg5 --no-swin > $O/pmc_${c}_stdout.txt 2>&1
done
F=$(find /tmp/p_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find /tmp/p_WRITE_SIZE -name "*counter_collection.csv" | head -1)
CALLS=$(python -c "import json;print(7 * json.loads([l for l in open('$O/pmc_FETCH_SIZE_stdout.txt') if l.startswith('{\"metric\"')][-1])['roofline']['gemm_calls_per_step'])")
python $R/tools/pmc_traffic.py "$F" "$W" gemm $O/pmc_traffic.json $CALLS > $O/pmc_traffic.txt 2>&1
tail -3 $O/pmc_traffic.txt
cd $R
cp $O/pmc_traffic.json profiles/r03_pmc_traffic.json
python bench.py > $O/bench_stdout.txt 2> $O/bench_stderr.txt
tail -1 $O/bench_stdout.txt > $O/bench.json
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['traffic'],d['roofline']['algorithmic_bytes_per_launch']);print(d['swin']['ms_per_step'],d['cbir']['ms_per_search'])"
