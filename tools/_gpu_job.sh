python -m pytest tests/test_attention.py tests/test_parity_bf16.py -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -3
python tools/bench_attention.py 2>&1 | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('form 3', d['short_sequence'], d['tflops']['short_sequence'])"
python bench.py --steps 20 --warmup 3 --no-parity --no-cbir --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('step',d['ms_per_step'],d['value'])"
