#!/bin/bash
# round 4: the native Swin engine on the MI355X -- GPU tests, the step at bs 128 both ways, kernel trace of the native step.  usage: r4_swin_job.sh [trace-only]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$1" != "trace-only" ]; then
  (cd $R && timeout 900 python -m pytest tests/test_swin.py -x -q -m gpu) > $O/swin_tests.log 2>&1; tail -3 $O/swin_tests.log
  timeout 300 python $R/tools/bench_swin.py 128 8 native > $O/swin_native.json 2> $O/swin_native.err; cat $O/swin_native.json
  timeout 300 python $R/tools/bench_swin.py 128 5 autograd > $O/swin_autograd.json 2> $O/swin_autograd.err; cat $O/swin_autograd.json
  timeout 300 python $R/tools/bench_swin.py 64 8 native > $O/swin_native64.json 2>/dev/null; cat $O/swin_native64.json
fi
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/swin_prof -o swin -- python $R/tools/bench_swin.py 128 5 native > $O/swin_prof.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/swin_prof -name "*.db" | head -1) > $O/swin_kernel_stats.txt
python $R/tools/rocpd_seq.py $(find /tmp/swin_prof -name "*.db" | head -1) > $O/swin_step_sequence.txt
head -3 $O/swin_step_sequence.txt
