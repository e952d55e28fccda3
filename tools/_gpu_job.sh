python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -5
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -5
