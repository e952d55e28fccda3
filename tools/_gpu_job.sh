mkdir -p gpurun_out/r2t
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pc -o c -- python $R/tools/cbir_pmc_run.py 8 > /tmp/pc.log 2>&1
cd $R
python tools/rocpd_stats.py $(find /tmp/pc -name "*.db" | head -1) | tee gpurun_out/r2t/cbir_kernel_stats.txt | head -16 | cut -c1-140
