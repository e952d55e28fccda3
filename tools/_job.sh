R=/root/repo; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d /tmp/pmc1 -o p --output-format csv -- python $R/tools/bench_swin.py 128 1 native > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/pmc2 -o p --output-format csv -- python $R/tools/bench_swin.py 128 1 native > $O/pmc2.log 2>&1
python $R/tools/pmc_sum.py window_attn $(find /tmp/pmc1 /tmp/pmc2 -name "*counter_collection.csv") > $O/wa_pmc.txt
cat $O/wa_pmc.txt
