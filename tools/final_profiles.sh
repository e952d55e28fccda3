#!/bin/bash
# Round-end measurement set on one MI355X box (run through gpurun): the bench line, the rocprofv3 kernel-trace summary of the same timed command, the PMC traffic
# passes of the GEMM (FETCH_SIZE / WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes) and of one CBIR search.  Everything lands in gpurun_out/final/;
# the summaries worth judging are copied into profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BARGS="--steps 20 --warmup 3 --no-parity --no-cpu-baseline --no-cbir --no-cfg5 --no-swin --no-cfg3"
rocprofv3 --kernel-trace --stats -d /tmp/p_trace -o t -- python $R/bench.py $BARGS > $O/trace_stdout.txt 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/p_trace -name "*.db" | head -1) > $O/bench_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/p_$c -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-parity --no-cpu-baseline --no-cbir --no-cfg5 --no-swin --no-cfg3 --no-other-operand > $O/pmc_${c}_stdout.txt 2>&1
done
F=$(find /tmp/p_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find /tmp/p_WRITE_SIZE -name "*counter_collection.csv" | head -1)
CALLS=$(python -c "import json;print(7 * json.loads([l for l in open('$O/pmc_FETCH_SIZE_stdout.txt') if l.startswith('{\"metric\"')][-1])['roofline']['dominant_kernel']['gemm_calls_per_step'])")
python $R/tools/pmc_traffic.py "$F" "$W" gemm $O/pmc_traffic.json $CALLS > $O/pmc_traffic.txt 2>&1
tail -3 $O/pmc_traffic.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/c_$c -o p --output-format csv -- python $R/tools/cbir_pmc_run.py 4 > $O/cbir_pmc_${c}_stdout.txt 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/c_SQ -o p --output-format csv -- python $R/tools/cbir_pmc_run.py 4 > $O/cbir_pmc_SQ_stdout.txt 2>&1
CF=$(find /tmp/c_FETCH_SIZE -name "*counter_collection.csv" | head -1); CW=$(find /tmp/c_WRITE_SIZE -name "*counter_collection.csv" | head -1); CS=$(find /tmp/c_SQ -name "*counter_collection.csv" | head -1)
python $R/tools/pmc_cbir.py "$CF" "$CW" "${CS:--}" 4 $O/cbir_pmc.json > $O/cbir_pmc.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/c_trace -o t -- python $R/tools/cbir_pmc_run.py 8 > $O/cbir_trace_stdout.txt 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/c_trace -name "*.db" | head -1) > $O/cbir_kernel_stats.txt
tail -8 $O/cbir_pmc.txt
cd $R
cp $O/pmc_traffic.json profiles/r06_pmc_traffic.json 2>/dev/null; cp $O/cbir_pmc.json profiles/r06_cbir_pmc.json 2>/dev/null    # the bench line below reads them
T0=$(date +%s); python bench.py > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "default bench.py wall: $(( $(date +%s) - T0 )) s" | tee $O/bench_wall.txt
tail -1 $O/bench_stdout.txt > $O/bench.json
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline'],d['cpu_baseline']);print(json.dumps(d['cbir'])[:1500])"
# the other measured configurations of SURVEY 8(d): swin_base (the shipped YAMLs' default backbone) through the native engine, and cfg3 (ConvNeXt-B + ArcFace over 1 M classes)
cd /tmp
python $R/tools/bench_swin.py 128 10 native > $O/swin_native.json 2>/dev/null; cat $O/swin_native.json
rocprofv3 --kernel-trace --stats -d /tmp/s_trace -o t -- python $R/tools/bench_swin.py 128 5 native > $O/swin_trace_stdout.txt 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/s_trace -name "*.db" | head -1) > $O/swin_kernel_stats.txt
python $R/tools/rocpd_seq.py $(find /tmp/s_trace -name "*.db" | head -1) > $O/swin_step_sequence.txt
python $R/tools/bench_cfg3.py 512 5 > $O/cfg3.json 2>/dev/null; cat $O/cfg3.json
rocprofv3 --kernel-trace --stats -d /tmp/f_trace -o t -- python $R/tools/bench_cfg3.py 512 3 > $O/cfg3_trace_stdout.txt 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/f_trace -name "*.db" | head -1) > $O/cfg3_kernel_stats.txt
# housekeeping re-timings on the current build: cfg1 (ResNet-18 plumbing config), the ConvNeXt-B classifier, cfg4's embedding extraction, the one-GPU overlap probe
python $R/tools/bench_cfg1.py 32 20 > $O/cfg1.json 2>/dev/null; cat $O/cfg1.json
python $R/tools/bench_cfg1.py 32 20 resnet18 fp16 > $O/cfg1_fp16.json 2>/dev/null; cat $O/cfg1_fp16.json      # the conforming-format arm (eager launches: the GradScaler step is not graph-captured)
python $R/tools/bench_cfg1.py 64 10 resnet50 > $O/resnet50.json 2>/dev/null; cut -c1-300 $O/resnet50.json
python $R/tools/bench_cfg4.py 2048 256 > $O/cfg4.json 2>/dev/null; cat $O/cfg4.json
python $R/tools/overlap_probe.py 256 150 2>/dev/null | grep '^{' > $O/overlap.json; cut -c1-300 $O/overlap.json
