mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_stdout.txt 2> gpurun_out/final/bench_stderr.txt
tail -1 gpurun_out/final/bench_stdout.txt > gpurun_out/final/bench.json
python -c "
import json;d=json.load(open('gpurun_out/final/bench.json'));c=d['cbir'];print(d['value'],d['ms_per_step'],d['roofline']['frac'],c['ms_per_search'],c['small_candidate_lists'])"
