#!/bin/bash
# Kernel traces of the headline step in both operand formats on one box (run through gpurun): gpurun_out/<dir>/{fp16,bf16}_kernel_stats.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-trace_operands}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for op in fp16 bf16; do
  rm -rf /tmp/p_$op
  rocprofv3 --kernel-trace --stats -d /tmp/p_$op -o t -- python $R/bench.py --operand $op --no-other-operand --steps 12 --warmup 3 --no-parity --no-cpu-baseline --no-cbir --no-cfg5 --no-swin > $O/${op}_stdout.txt 2>&1
  python $R/tools/rocpd_stats.py $(find /tmp/p_$op -name "*.db" | head -1) > $O/${op}_kernel_stats.txt
done
