for f in 3 2 1; do
VDK_ATTN_BWD_FORM=$f python bench.py --steps 20 --warmup 3 --no-parity --no-cbir --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('form $f',d['ms_per_step'],d['value'])"
done
