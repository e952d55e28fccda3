#!/bin/bash
# PMC passes over one GEMM shape for both 256x256 structures.  usage: w4_pmc.sh OUTDIR M N K
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/$1; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for kern in 2 5; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pa_$kern -o p --output-format csv -- python $R/tools/w4_pmc_run.py $kern $2 $3 $4 > $O/pa_$kern.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d /tmp/pb_$kern -o p --output-format csv -- python $R/tools/w4_pmc_run.py $kern $2 $3 $4 > $O/pb_$kern.log 2>&1
  python $R/tools/pmc_sum.py gemm $(find /tmp/pa_$kern /tmp/pb_$kern -name "*counter_collection.csv") > $O/pmc_kern$kern.txt 2>&1
  cat $O/pmc_kern$kern.txt
done
