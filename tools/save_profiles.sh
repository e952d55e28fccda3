#!/bin/bash
# copy the round's judged summaries from gpurun_out/final (scratch, merged back by gpurun) into profiles/ (tracked)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/final; P=$R/profiles; N=${1:-r06}
cp $O/bench.json $P/${N}_bench.json
cp $O/bench_kernel_stats.txt $P/${N}_bench_kernel_stats.txt
cp $O/pmc_traffic.json $P/${N}_pmc_traffic.json; cp $O/pmc_traffic.txt $P/${N}_pmc_traffic.txt
cp $O/cbir_pmc.json $P/${N}_cbir_pmc.json; cp $O/cbir_pmc.txt $P/${N}_cbir_pmc.txt; cp $O/cbir_kernel_stats.txt $P/${N}_cbir_kernel_stats.txt
cp $O/swin_native.json $P/${N}_swin_native.json; cp $O/swin_kernel_stats.txt $P/${N}_swin_kernel_stats.txt; cp $O/swin_step_sequence.txt $P/${N}_swin_step_sequence.txt
cp $O/cfg3.json $P/${N}_cfg3.json; cp $O/cfg3_kernel_stats.txt $P/${N}_cfg3_kernel_stats.txt
cp $O/cfg1.json $P/${N}_cfg1_resnet18.json; cp $O/cfg1_fp16.json $P/${N}_cfg1_resnet18_fp16.json; cp $O/resnet50.json $P/${N}_resnet50_bs64.json; cp $O/cfg4.json $P/${N}_cfg4_convnext_embeddings.json; cp $O/overlap.json $P/${N}_overlap_1gpu.json
cp $O/bench_wall.txt $P/${N}_bench_wall.txt
ls -la $P | grep ${N}_ | wc -l
