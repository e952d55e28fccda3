#!/bin/bash
# CBIR A/B on one box: approximate stage ranking on / off, geometric stages on / off (ms per search, checksums must agree), the kernel trace of the default search, GPU parity tests.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r4_cbir}
mkdir -p $O
cd $R
for rep in 1 2; do for a in 1 0; do for g in 1 0; do VDK_CBIR_APPROX_RANK=$a VDK_CBIR_GEO=$g python tools/cbir_quick.py default 10 2>/dev/null | tee -a $O/cbir_ab.txt; done; done; done
python -m pytest tests/test_cbir.py -m gpu -q 2>&1 | tail -3 | tee $O/cbir_tests.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c_trace
rocprofv3 --kernel-trace --stats -d /tmp/c_trace -o t -- python $R/tools/cbir_pmc_run.py 8 > $O/cbir_trace_stdout.txt 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/c_trace -name "*.db" | head -1) > $O/cbir_kernel_stats.txt
head -8 $O/cbir_kernel_stats.txt
