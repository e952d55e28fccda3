cd /root/repo
timeout 900 python -m pytest tests/test_swin.py -x -q -m gpu -s 2>&1 | grep -E "passed|failed|logits|Error" | tail -6
for op in fp16 bf16 fp16 bf16; do timeout 300 python tools/bench_swin.py 128 10 native $op 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['operand'], d['ms_per_step'], d['losses'])"; done
