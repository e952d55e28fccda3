mkdir -p gpurun_out/r2d
python -m pytest tests -m gpu -q > gpurun_out/r2d/pytest.log 2>&1; grep -E "^FAILED|passed|failed|^E  " gpurun_out/r2d/pytest.log | cut -c1-400 | tail -20
python tools/attn_ablate.py > gpurun_out/r2d/ablate.json 2> gpurun_out/r2d/ablate.err; cat gpurun_out/r2d/ablate.json
python tools/bench_attention.py > gpurun_out/r2d/attn.json 2>/dev/null; cat gpurun_out/r2d/attn.json
python bench.py > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r2d/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print(d.get('parity')); print(d['cpu_baseline']); print(d['cbir'].get('cpu_baseline_torch_topk'))"; tail -3 gpurun_out/r2d/bench.err
