for i in 1 2; do
python bench.py --steps 20 --warmup 3 --no-parity --no-cbir --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('events',d['ms_per_step'],d['value'])"
VDK_BENCH_NO_EVENTS=1 python bench.py --steps 20 --warmup 3 --no-parity --no-cbir --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('no events',d['ms_per_step'],d['value'])"
done
