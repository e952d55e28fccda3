#!/bin/bash
# Round-4 GEMM measurement set on one MI355X box (run through gpurun): the GEMM table with the vendor column for both operand formats, and the effective clock per kernel
# of the headline step (one PMC pass, GRBM_GUI_ACTIVE, kernel trace only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r4_gemm}
mkdir -p $O
cd $R
python tools/bench_gemm_w4.py $O/gemm_ab_bf16.json 3 - bf16 > $O/gemm_ab_bf16.txt 2>&1
python tools/bench_gemm_w4.py $O/gemm_ab_fp16.json 3 - fp16 > $O/gemm_ab_fp16.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for op in fp16 bf16; do
  rm -rf /tmp/c_$op
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d /tmp/c_$op -o p --output-format csv -- python $R/bench.py --operand $op --no-other-operand --steps 4 --warmup 2 --no-parity --no-cpu-baseline --no-cbir --no-cfg5 --no-swin > $O/clock_${op}_stdout.txt 2>&1
  F=$(find /tmp/c_$op -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_clock.py "$F" $O/clock_$op.json > $O/clock_$op.txt 2>&1
  head -3 "$F" > $O/clock_${op}_csv_head.txt
done
tail -12 $O/clock_fp16.txt
