mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_stdout.txt 2> gpurun_out/final/bench_stderr.txt
tail -1 gpurun_out/final/bench_stdout.txt > gpurun_out/final/bench.json
python -c "
import json;d=json.load(open('gpurun_out/final/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac']);c=d['cbir'];print(c['ms_per_step'] if 'ms_per_step' in c else c['ms_per_search'], c['float16_storage']['ms_per_search'], c['d512']['ms_per_search'], c['optimistic_two_stage_schedule']['ms_per_search'])"
